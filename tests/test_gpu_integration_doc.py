"""The binding example in INTEGRATION.md is executed as written (against libgn2v.so and a stand-in
for the ensmallen.Graph getters it calls), so the document cannot drift from the ABI."""
import os
import re

import numpy as np
import pytest

import embiggen_amd as E
from embiggen_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class EnsmallenLikeGraph:
    def __init__(self, g):
        self.g = g

    def get_number_of_nodes(self): return self.g.get_number_of_nodes()
    def get_cumulative_node_degrees(self): return self.g.row_ptr[1:]
    def get_directed_destination_node_ids(self): return self.g.col_idx
    def is_directed(self): return self.g.is_directed()


def test_the_documented_binding_runs_and_matches_the_package():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# embiggen/embedders/amd_embedders/gn2v_models.py.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libgn2v.so")', f'C.CDLL({_lib.build()!r})')
    _lib.lib()  # torch's HIP runtime first, as the package does
    scope = {}
    exec(compile(block, "INTEGRATION.md", "exec"), scope)
    graph = E.karate_club()
    kw = dict(epochs=2, walk_length=16, iterations=2, window_size=3, number_of_negative_samples=4)
    central, contextual = scope["SkipGram"](8, 42, **kw).fit_transform(EnsmallenLikeGraph(graph))
    assert central.shape == (34, 8) and central.dtype == np.float32 and np.isfinite(central).all()
    # same engine, same seed, same (atomic) schedule family: statistically the package's result
    ours = E.models.SkipGram(embedding_size=8, random_state=42, verbose=False, **kw).fit_transform(graph)
    assert np.abs(central - ours[0]).max() < 0.05 and np.abs(contextual - ours[1]).max() < 0.05
    # struct layouts of the document == the package's
    import ctypes as C
    assert C.sizeof(scope["WalkParams"]) == C.sizeof(_lib.WalkParams)
    assert C.sizeof(scope["TrainParams"]) == C.sizeof(_lib.TrainParams)
    assert [f[0] for f in scope["WalkParams"]._fields_] == [f[0] for f in _lib.WalkParams._fields_]
    assert [f[0] for f in scope["TrainParams"]._fields_] == [f[0] for f in _lib.TrainParams._fields_]


def test_whole_fit_from_a_plain_c_program(tmp_path):
    """tests/c/fit_from_c.c: graph, tables (hipMalloc) and `gn2v_train` for SkipGram and CBOW from a
    C11 program with neither Python nor PyTorch in the process; it checks the stats and that the
    embedding separates the cliques of its graph."""
    import shutil
    import subprocess

    from embiggen_amd import _lib

    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    lib_dir = os.path.dirname(_lib.build())
    exe = str(tmp_path / "fit_from_c")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__",
                    "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "c", "fit_from_c.c"),
                    "-L", lib_dir, "-lgn2v", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.stdout, res.stderr)
    lines = res.stdout.strip().splitlines()
    assert lines[-1] == "ok" and lines[0].startswith("model 0") and lines[1].startswith("model 1")


def test_multi_gpu_fit_from_a_plain_c_program_over_rccl(tmp_path):
    """tests/c/fit_world_rccl.c: gn2v_train_world with the communicator of include/gn2v_rccl.h --
    RCCL loaded by libgn2v.so itself, neither Python nor PyTorch in the process.  One GPU here:
    world = 1 (RCCL initialised, the ranks' agreement all-gathered through it); the same program
    is what a launcher starts once per GPU."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    lib_dir = os.path.dirname(_lib.build())
    exe = str(tmp_path / "fit_world_rccl")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__",
                    "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "c", "fit_world_rccl.c"),
                    "-L", lib_dir, "-lgn2v", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    res = subprocess.run([exe, "0", "1", str(tmp_path / "job.id")], capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, (res.stdout, res.stderr)
    lines = res.stdout.strip().splitlines()
    # (RCCL prints its version banner first)
    assert lines[-1] == "ok" and any(ln.startswith("rank 0 of 1: pairs ") for ln in lines), lines
    assert not (tmp_path / "job.id").exists()  # rank 0 removes the id file at the end


def test_the_readme_quick_start_runs_as_written():
    text = open(os.path.join(ROOT, "README.md")).read()
    block = re.search(r"```python\n(import embiggen_amd as E\ngraph = .*?)```", text, re.S).group(1)
    scope = {}
    exec(compile(block, "README.md", "exec"), scope)  # noqa: S102
    central, contextual = scope["central"], scope["contextual"]
    assert central.shape == contextual.shape == (34, 128) and central.dtype == np.float32
    assert np.isfinite(central).all() and np.isfinite(contextual).all()
