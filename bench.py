"""Headline benchmark: SkipGram training-pairs/sec (+ walk-steps/sec), d = 128, on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` (N > 1) outside a torch.distributed job starts that job itself: the
parent process -- which never touches HIP or torch -- runs the second line as a child process
with `--master-addr 127.0.0.1` and a free port, relays rank 0's JSON line and exits with the
job's status.

Workload (the graph BASELINE.json's target is quoted on): seeded
Barabasi-Albert graph, 10 M nodes / 100 M edges, Node2Vec SkipGram with the reference's default
hyper-parameters (walk_length 128, window 5, 10 negatives, return_weight 0.25, explore_weight 4;
embiggen/embedders/ensmallen_embedders/node2vec_skipgram.py:11-24) at d = 128, f32.
One step = one pass of the hot path over one batch of synthetic input: generate `--walks`
second-order walks on the GPU and train on all of them (1 250 pairs per walk).  Graph and both
tables are resident in HBM before the timed region.

N > 1: one process per GPU, CSR replicated, every rank generates its own slice of the walk ids
(weak scaling: per-GPU work fixed).  The central table is striped over the ranks (node id % N), the
contextual table over P N parts (P >= 2 per rank) that travel round the ranks: in every episode a GPU trains the
pairs (centre it owns, context in the resident part) with negatives from that part while the part
it just finished and the part it needs next are in flight (RCCL send / receive).  The round's walks
are all-gathered (512 B per walk) and every rank extracts and sorts its own pairs on the device,
on a second stream while the previous round trains.  No row is ever shared, so N GPUs compute
exactly what the single-process simulation of the tests computes.  See DESIGN.md "Multi-GPU".

`--model cbow` times the CBOW kernel on the same workload (unit: centres/s; with N > 1 as N
independent replicas: CBOW does not shard).  A/B switches: `--central-store`
(a centre's only run in a cell stores row + gradient instead of adding with atomics), `--reserve-cus k` (training kernel on a CU-masked stream
that leaves k CUs of every XCD to RCCL), `--record`, `--group-parts`, `--round-walks`,
`--overlap`.  Measurement aids, one GPU:
`--phantom-world N` runs one rank of an N-GPU job with its true geometry and no fabric (the line
says so and is not the benchmark's value); `--stripes V` the optional centre stripes of
DESIGN.md 7.4; GN2V_BENCH_MEMLOG=1 logs the allocator's state around the phases.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_PAIR = 12288  # 2 * (k + 2) * d * 4 with k = 10, d = 128
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# the same guide: 256 CUs x 4 SIMDs, 2 400 MHz max clock, a wave64 VALU instruction occupies its
# SIMD for 4 cycles: 614.4 G wave-instructions/s is all the vector pipes can issue
VALU_ISSUE_PEAK_GIPS = 256 * 4 * 2.4e9 / 4 / 1e9
# f32 atomic adds the chip's L2 / memory-side atomic units retire per second, whatever the shape
# of the instruction and whether the rows live in L2 or in HBM: scripts/atomic_probe.hip measured
# 3.13e11 (5 GB table) and 3.31e11 (51 MB table) dword adds/s on an MI355X
# (profiles/r05_logs/r5_atomic_probe.log) -- one dword per clock for each of 128 L2 channels at
# 2.4-2.6 GHz; the guide quotes no figure for atomics.  The higher one is the ceiling.
L2_ATOMIC_PEAK_GDWORDS = 331.0


def valu_floor_per_pair(ld, k):
    """Vector instructions (per wave) the arithmetic of ONE training pair needs at the very
    least in the resident kernel's layout: a wave scores four samples at a time
    (one per 16-lane group, 4 row floats x ld / 64 per lane), and a sample is: dot product
    2 ld / 64 packed FMAs + 4 DPP adds of the 16-lane reduction; clamp, exp2 scale, exp, + 1,
    reciprocal, label - sigma, x learning rate = 6; gradient 2 ld / 64 packed FMAs; row update
    2 ld / 64 packed FMAs.  (k + 1) samples per pair, four pairs side by side."""
    ch = max(1, -(-ld // 64))
    return (k + 1) * (6 * ch + 10) / 4.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--walks", type=int, default=1 << 20, help="walks per step per GPU")
    ap.add_argument("--batch", type=int, default=1 << 16, help="walks per training launch")
    ap.add_argument("--walk-batch", type=int, default=1 << 19,
                    help="walks per walk-kernel launch (the sampler is latency bound: it needs "
                         "many more walkers in flight than one training launch consumes)")
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--mode", default="auto",
                    choices=["auto", "write_through", "write_back", "atomic"])
    ap.add_argument("--return-weight", type=float, default=0.25)
    ap.add_argument("--explore-weight", type=float, default=4.0)
    ap.add_argument("--max-neighbours", type=int, default=100,
                    help="the reference's default (node2vec_skipgram.py:22): steps out of nodes of "
                         "higher degree choose among a per-visit sub-sample of that many edges; "
                         "0 = None = exact walks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calibrate", action="store_true",
                    help="run the traffic-calibration kernel instead of training (for rocprofv3 "
                         "--pmc passes): every table row touched exactly once per launch")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--parallelism", default="auto", choices=["auto", "single", "blocks"],
                    help="blocks (= auto): the block trainer, on one GPU too (contextual rows in "
                         "XCD-exclusive cells); single: the walk-ordered kernel, 1 GPU only")
    ap.add_argument("--round-walks", type=int, default=0,
                    help="blocks: walks per rank per round (every context part visits every rank "
                         "once per round; a round may span several steps; the longer, the more "
                         "pairs of a centre meet in a cell: kernel 0.82 / 0.92 / 0.96 of the "
                         "roofline at 2^20 / 2^22 / 2^23).  0 = gn2v_block_round_plan: 2^23 on the "
                         "bench graph")
    ap.add_argument("--group-parts", type=int, default=0,
                    help="blocks: parts whose pairs are extracted, sorted and held at a time "
                         "(8 B per pair; 0 = gn2v_block_round_plan: at least four groups per round)")
    ap.add_argument("--parts", type=int, default=None,
                    help="blocks: context parts (default 1 on one GPU, 2 x world otherwise)")
    ap.add_argument("--slices", type=int, default=None,
                    help="blocks: XCD slices inside a part (8 = every XCD owns its rows; default: "
                         "distributed.auto_plan)")
    ap.add_argument("--record", type=int, default=32, help="blocks: pairs per record")
    ap.add_argument("--hot-rows", type=int, default=None,
                    help="blocks: rows of every cell (highest in-degrees) whose updates accumulate "
                         "in LDS and reach the row through atomics (default: the library's; 0 = "
                         "none, plain stores on every contextual row)")
    ap.add_argument("--hot-flush", type=int, default=0,
                    help="blocks: a hot row's pending sum is handed over on average every this "
                         "many updates per workgroup (power of two; 0 = 16)")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="blocks: CUs of every XCD left to other work (RCCL's transfer kernels): "
                         "the training kernel runs on a CU-masked stream (gn2v_graph_reserve_cus)")
    ap.add_argument("--central-store", action="store_true",
                    help="blocks: a run that is its centre's only one in the cell stores row + "
                         "gradient (round 3's default; loses the update when another XCD holds "
                         "the centre at that moment) instead of adding the gradient with atomics")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off"],
                    help="blocks: prepare round t + 1 on a second stream while round t trains "
                         "(auto: with several GPUs, where it hides the walk all-gather; on one GPU "
                         "preparation and training share the same HBM and nothing is gained)")
    ap.add_argument("--stripes", type=int, default=0,
                    help="blocks on one GPU: centre stripes trained one after the other over the "
                         "pairs of a round of `stripes` x as many walks -- the memory of a round "
                         "of --round-walks / stripes walks, centre runs `stripes` times as long; "
                         "faster (8 stripes: +7 %%) but the stripes of a round are trained one "
                         "after the other, which costs link quality: 0 = 1 = off, what ships")
    ap.add_argument("--job-timeout", type=float, default=1500.0,
                    help="--gpus N > 1 started from a plain shell: seconds after which the parent "
                         "terminates the job's process group and prints a JSON line with "
                         "\"error\": \"timeout\", the phase each rank had reached and the last stderr "
                         "lines (0 = no watchdog)")
    ap.add_argument("--entry", default="auto", choices=["auto", "c", "python"],
                    help="blocks on one GPU: which host loop drives the rounds.  c (= auto when no "
                         "plan switch is given): ONE call of gn2v_train, the C-ABI entry "
                         "INTEGRATION.md binds in place of the reference's "
                         "self._model.fit_transform(graph) (node2vec.py:99), with the steps' walks "
                         "as its walk budget; python: embiggen_amd.distributed's trainer (the "
                         "multi-GPU host loop, which every N > 1 uses)")
    ap.add_argument("--model", default="skipgram", choices=["skipgram", "cbow"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="testing only: all ranks use GPU 0 (use with --backend gloo)")
    ap.add_argument("--phantom-world", type=int, default=0,
                    help="measurement aid, --gpus 1 only: run ONE rank of a world of this size "
                         "with its true geometry (central stripe, travelling parts, pairs extracted "
                         "from the walks of all ranks, two rounds in flight) and no fabric: the "
                         "other ranks' walks are generated locally, a hop is a local copy.  Gives a "
                         "rank's compute rate and HBM footprint at that world size; the line says "
                         "'phantom' and is not the benchmark's value")
    ap.add_argument("--phantom-rank", type=int, default=0)
    return ap.parse_args()


class PhantomComm:
    """One rank of a world of `world` ranks without peers (--phantom-world): the caller hands over
    the walks of all ranks already concatenated, a part that leaves comes back as the part that
    arrives (same size up to one row; the values are those of a trained part).  Like RCCL's, the
    copy runs on a stream of its own, ordered after the training launched so far and awaited by
    the compute stream after the next launch; HIP events around it tell how long a small kernel
    that arrives while the training kernel occupies the CUs takes to get through."""

    backend = "phantom"

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self._stream, self.hops = None, []

    def all_gather(self, tensor):
        return tensor

    def sendrecv_start(self, send, dst, recv, src):
        import torch

        if self._stream is None:
            self._stream = torch.cuda.Stream()
        main, side = torch.cuda.current_stream(), self._stream
        side.wait_stream(main)
        rows = min(send.shape[0], recv.shape[0])
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            t0.record()
            recv[:rows].copy_(send[:rows])
            if recv.shape[0] > rows:
                recv[rows:].copy_(send[: recv.shape[0] - rows])
            t1.record()
        self.hops.append((t0, t1))

        class _Hop:
            def wait(_self):
                main.wait_stream(side)

        return _Hop()

    def hop_ms(self):
        """(count, mean, max) of the hop copies' durations so far; clears the list."""
        import torch

        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.hops]
        self.hops = []
        return (len(ms), sum(ms) / max(len(ms), 1), max(ms, default=0.0))


def usable_cores() -> int:
    """Host cores this process may actually use: min(logical CPUs, affinity mask, cgroup quota)."""
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def committed_traffic(traffic_key):
    """((HBM bytes per algorithmic byte, file), None) of the dominant kernel from the committed
    rocprofv3 --pmc passes (profiles/*_pmc.json, made by scripts/profile_bench.sh +
    summarize_profiles.py) when they were taken on this very workload and schedule
    (``config.traffic_key``); (None, reason) otherwise -- counters cannot be read from inside the
    run, and a line without its traffic figure must say why."""
    import glob

    best, seen = None, []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        key = rec.get("config", {}).get("traffic_key")
        seen.append(f"{os.path.basename(path)}: {key}")
        if key == traffic_key:
            # per algorithmic byte, so that launches of another size (a run whose steps do not
            # fill whole rounds) are priced by what they process
            best = (rec["hbm_traffic_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"],
                    os.path.basename(path))
    if best is not None:
        return best, None
    return None, (f"no committed profiles/*_pmc.json was taken on traffic_key {traffic_key!r} "
                  f"(run scripts/profile_bench.sh on this command line); found: {seen}")


def committed_counters(kernel, ld, kernel_pairs_per_s):
    """What the committed rocprofv3 counter passes of the resident kernel say about its two
    bounds (profiles/r06_atomic_counters.json: TCC counters of the kernel and of the atomic probe;
    profiles/r06_resident_counters.json: SQ instruction counters): `atomic_unit_busy` -- the
    TCCs' busy share and the atomic sectors they retire per channel-cycle next to the probe's --
    and `valu_issued_frac`, the vector instructions REALLY issued per pair x this run's kernel
    pairs/s over the chip's issue rate (valu_issue_frac prices the arithmetic's floor only)."""
    out = {}
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r06_atomic_counters.json")))
        mine = next(v for k, v in rec["bench"].items() if kernel.split("::")[-1] in k)
        probe = max(v["atomic_sectors_per_channel_cycle"]
                    for k, v in rec["atomic_probe_10000000_rows"].items()
                    if v.get("atomic_sectors_per_channel_cycle"))
        out["atomic_unit_busy"] = {
            "tcc_busy": mine["tcc_busy"],
            "atomic_sectors_per_channel_cycle": mine["atomic_sectors_per_channel_cycle"],
            "probe_atomic_sectors_per_channel_cycle": probe,
            "of_probe_per_cycle": mine["atomic_sectors_per_channel_cycle"] / probe,
            "source": "profiles/r06_atomic_counters.json (rocprofv3 --pmc TCC_ATOMIC_SECTORS_sum "
                      "TCC_BUSY_sum TCC_CYCLE_sum of this kernel and of scripts/atomic_probe.hip; "
                      "profiles/r06_atomic_summary.md)"}
    except (OSError, ValueError, KeyError, StopIteration):
        out["atomic_unit_busy"] = None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r06_resident_counters.json")))
        if rec["bench"]["kernel"] == kernel and ld == 128:
            issued = rec["per_pair"]["SQ_INSTS_VALU"]
            out["valu_issued_per_pair"] = issued
            out["valu_issued_frac"] = issued * kernel_pairs_per_s / 1e9 / VALU_ISSUE_PEAK_GIPS
            out["valu_issued_source"] = "profiles/r06_resident_counters.json (SQ_INSTS_VALU per pair)"
    except (OSError, ValueError, KeyError):
        pass
    return out


def cpu_baseline(graph, args, central, contextual, seconds):
    """The oracle's OpenMP Hogwild restatement timed on this box's host cores, on a bounded
    sample of the same workload (same graph, same parameters, walk ids past the GPU's)."""
    import numpy as np

    from oracle import oracle as O

    cores = usable_cores()
    # the tuned build is compiled for THIS host (-march=native), not for wherever the tree came from
    O.build(force=True, fast=True)
    og = O.OracleGraph(graph.row_ptr, graph.col_idx)
    d = args.d
    c = central[:, :d].contiguous().cpu().numpy()
    x = contextual[:, :d].contiguous().cpu().numpy()
    wp = O.WalkParams(128, 10, args.return_weight, args.explore_weight, args.max_neighbours, 0)
    cbow = args.model == "cbow"
    tp = O.TrainParams(1 if cbow else 0, d, d, 1, 10, 5, 0.01, 0.9, 6.0, O.FLAG_SCALE_FREE,
                       d ** -0.5)

    def run(first, n):
        t0 = time.perf_counter()
        w = O.walks(og, wp, 42, 0, first, n)
        t1 = time.perf_counter()
        # the tuned build of the oracle's Hogwild trainer (vectorised dot products, software
        # prefetch of the next context slot's rows): 3-4 x the strict build the tests check with
        O.train_walks(og, tp, w, 42, 0, first, 0.01, c, x, threads=cores, fast=True)
        t2 = time.perf_counter()
        # the metric's unit: (centre, context) pairs for SkipGram, centres for CBOW
        pairs = int(n) * (128 if cbow else 2 * 5 * 128 - 5 * 6)
        return pairs, n * 127, t1 - t0, t2 - t1

    first = 1 << 40  # walk ids the GPU run never uses
    probe_n = 64 * cores
    pairs, _, _, tt = run(first, probe_n)
    rate = pairs / max(tt, 1e-6)
    n = int(max(probe_n, min(seconds * rate / (128 if cbow else 1250), 1 << 20)))
    pairs, steps, tw, tt = run(first + probe_n, n)
    return {
        "value": pairs / (tw + tt),
        "unit": "centres/s" if cbow else "pairs/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} walks ({pairs} {'centres' if cbow else 'pairs'}) of the same BA graph and parameters, walks+training "
                  f"{tw + tt:.1f}s, OpenMP Hogwild oracle (tuned build: -ffast-math, software "
                  f"prefetch) on {cores} threads",
        "train_only_per_s": pairs / tt,
        "walk_steps_per_s": steps / max(tw, 1e-9),
    }


PHASE_TAG = "[bench phase] "


def phase(name):
    """Workers mark where they are (stderr, every rank): the parent's watchdog reports the last
    mark when a job hangs.  GN2V_BENCH_STALL=<phase>:<seconds> makes a worker sleep on reaching
    that phase (tests: a stand-in for an RCCL bootstrap that never returns)."""
    print(f"{PHASE_TAG}{name} rank={os.environ.get('RANK', '0')} t={time.time():.3f}",
          file=sys.stderr, flush=True)
    stall = os.environ.get("GN2V_BENCH_STALL", "")
    if stall.partition(":")[0] == name:
        time.sleep(float(stall.partition(":")[2] or 1e9))


def launch_job(args):
    """`bench.py --gpus N` outside a torch.distributed job: run it as the job the contract names
    (one process per GPU), relay rank 0's JSON line, return the job's status.  This process never
    initialises HIP, torch.cuda or RCCL: the workers are fresh children, nothing is re-exec'd.
    A watchdog bounds the job: after --job-timeout seconds the whole process group of the job is
    terminated and ONE JSON line says so -- error, the phase every rank had reached and the last
    lines of the job's stderr -- so that a hang (an RCCL bootstrap that never completes) leaves a
    record instead of an expired lease."""
    import signal
    import socket
    import subprocess
    import threading

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    # a session of its own: the watchdog can end the launcher AND its workers as one group
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                            text=True, start_new_session=True)
    state = {"json": None, "tail": [], "phases": {}}

    def read_stdout():
        for out_line in proc.stdout:
            if out_line.lstrip().startswith("{"):
                state["json"] = out_line.strip()
            else:
                sys.stderr.write(out_line)  # RCCL banners and the like

    def read_stderr():
        for err_line in proc.stderr:
            sys.stderr.write(err_line)
            if err_line.startswith(PHASE_TAG):
                fields = err_line[len(PHASE_TAG):].split()
                rank = next((f[5:] for f in fields if f.startswith("rank=")), "?")
                state["phases"][rank] = fields[0]
            else:
                state["tail"] = (state["tail"] + [err_line.rstrip()])[-20:]

    readers = [threading.Thread(target=read_stdout, daemon=True),
               threading.Thread(target=read_stderr, daemon=True)]
    for t in readers:
        t.start()
    try:
        status = proc.wait(timeout=args.job_timeout if args.job_timeout > 0 else None)
    except subprocess.TimeoutExpired:
        for sig, grace in ((signal.SIGTERM, 10), (signal.SIGKILL, 10)):
            try:
                os.killpg(proc.pid, sig)  # the job's own process group, nothing else
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        for t in readers:
            t.join(timeout=5)
        print(json.dumps({
            "error": "timeout",
            "detail": f"the {args.gpus}-process job did not finish within --job-timeout "
                      f"{args.job_timeout} s and was terminated",
            "phase_reached_per_rank": state["phases"],
            "stderr_tail": state["tail"],
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "argv": sys.argv[1:]}), flush=True)
        return 124
    for t in readers:
        t.join(timeout=5)
    if state["json"] is not None:
        print(state["json"], flush=True)
    return status


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_job(args))
    import torch

    import embiggen_amd as E
    from embiggen_amd import _lib, ops

    phase("start")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the job was launched with a "
                         "different number of processes than --gpus says")
    _lib.require_device()
    if args.share_device:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        phase("init_process_group")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=args.backend)

    cbow = args.model == "cbow"
    phase("graph")
    graph = E.barabasi_albert(args.nodes, args.m, 42, device=local)
    n, d = graph.get_number_of_nodes(), args.d
    reserved = ops.graph_reserve_cus(graph, args.reserve_cus, local) if args.reserve_cus else None
    # the engine's padded row stride (models.SkipGram.padded_size)
    ld = (d + 3) // 4 * 4 if d <= 16 else (d + 31) // 32 * 32 if d <= 128 else (d + 63) // 64 * 64
    flags = _lib.TRAIN_SCALE_FREE | {
        "auto": 0, "write_through": _lib.TRAIN_WRITE_THROUGH,
        "write_back": _lib.TRAIN_WRITE_BACK, "atomic": _lib.TRAIN_ATOMIC}[args.mode]
    if args.central_store:
        flags |= _lib.TRAIN_CENTRAL_STORE
    if args.calibrate:
        central = ops.init_table(n, d, 42, 0, d ** -0.5, device=local, ld=ld)
        perm = torch.randperm(n, device="cuda", dtype=torch.int64).to(torch.int32)
        mode_flags = flags & ~_lib.TRAIN_SCALE_FREE
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = args.warmup + args.steps
        for _ in range(reps):
            ops.touch_rows(central, perm, mode_flags)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        row_bytes = central.shape[1] * 4
        print(json.dumps({"calibration": "gn2v::touch_rows_kernel", "rows_per_launch": n,
                          "read_bytes_per_launch": n * row_bytes + n * 4,
                          "write_bytes_per_launch": n * row_bytes, "update_mode": args.mode,
                          "ms_per_launch": dt * 1e3,
                          "GBps": (2 * n * row_bytes + 4 * n) / dt / 1e9}), flush=True)
        return
    model_id = _lib.MODEL_CBOW if cbow else _lib.MODEL_SKIPGRAM
    tp = ops.train_params(model_id, d, 10, 5, lr=0.01, flags=flags, ld=ld)
    wp = ops.walk_params(128, 10, args.return_weight, args.explore_weight,
                         args.max_neighbours or None)
    from embiggen_amd.distributed import (BlockPartitionedTrainer, LoopbackComm, TorchComm,
                                          walk_slice)

    mode = args.parallelism
    if mode == "auto":
        mode = "single" if cbow else "blocks"
    if mode == "single" and world > 1 and not cbow:
        raise SystemExit("--parallelism single needs --gpus 1")
    replicas = mode == "single" and world > 1  # CBOW does not shard: N independent fits
    if mode == "blocks" and cbow:
        raise SystemExit("the block-partitioned trainer is SkipGram only")
    phantom = args.phantom_world > 1
    if phantom and (world > 1 or mode != "blocks"):
        raise SystemExit("--phantom-world needs --gpus 1 and the block trainer")
    blocks = comm = None
    # N = 1, nothing but the defaults: the timed region is one call of the C-ABI entry
    plan_switches = (args.parts is not None or args.slices is not None or args.record != 32
                     or args.hot_rows is not None or args.hot_flush or args.group_parts
                     or args.overlap == "on" or args.reserve_cus)
    c_entry = (mode == "blocks" and world == 1 and not phantom
               and (args.entry == "c" or (args.entry == "auto" and not plan_switches)))
    if args.entry == "c" and not c_entry:
        raise SystemExit("--entry c needs --gpus 1, the block path and no --phantom-world")
    overlap = args.overlap == "on" or (args.overlap == "auto" and (world > 1 or phantom))
    # the trainer's view of the job (a phantom rank sees the world it stands in for)
    t_rank, t_world = (args.phantom_rank, args.phantom_world) if phantom else (rank, world)
    stripes = 1
    phase("trainer")
    if c_entry:
        central = torch.empty((n, ld), dtype=torch.float32, device=f"cuda:{local}")
        contextual = torch.empty((n, ld), dtype=torch.float32, device=f"cuda:{local}")
        c_stats = _lib.Stats()
    elif mode == "blocks":
        # tables partitioned by node id; no row is ever held by two GPUs
        comm = (PhantomComm(t_rank, t_world) if phantom
                else TorchComm() if world > 1 else LoopbackComm())
        stripes = args.stripes if args.stripes and world == 1 and not phantom else 1
        blocks = BlockPartitionedTrainer(graph, tp, d, ld, 42, d ** -0.5, comm, f"cuda:{local}",
                                         walk_length=128, window=5, parts=args.parts,
                                         slices=args.slices, record=args.record,
                                         hot_rows=args.hot_rows, hot_flush=args.hot_flush,
                                         stripes=stripes)
        if overlap:  # --overlap on, one GPU: the several-rank loop with its side stream
            blocks.round_driver = False
    else:
        central = ops.init_table(n, d, 42, 0, d ** -0.5, device=local, ld=ld)
        contextual = ops.init_table(n, d, 42, 1, d ** -0.5, device=local, ld=ld)
        # gn2v_train allocates its round buffers with hipMalloc: what building the graph left in
        # torch's caching allocator goes back to the driver first
        torch.cuda.empty_cache()

    if blocks is not None:
        from embiggen_amd.distributed import round_plan

        torch.cuda.empty_cache()  # what building the graph left in the allocator's cache
        # walks one pass extracts from (times the stripes that share them) and the parts whose
        # pairs are held at a time, from what is free now
        cap = args.round_walks // max(1, stripes)
        if not args.round_walks and getattr(blocks, "permute", False):
            # resident cells: at least 16 rounds per epoch of the graph (10 walks a node),
            # the rule of gn2v_train_blocks / models.fit_transform_blocks
            from embiggen_amd.distributed import rounds_per_epoch

            shortest = int(os.environ.get("GN2V_ROUND_MIN_WALKS", "")
                           or (1 << 19 if t_world > 1 else 1 << 14))
            cap = max(shortest, -(-n * 10 // (rounds_per_epoch(1) * max(t_world, stripes))))
        # (the plan is made for the round that will be trained: its cap)
        auto_walks, auto_group = round_plan(
            torch.cuda.mem_get_info()[0], n, 128, 5, max(t_world, stripes), blocks.parts,
            blocks.slices, overlap and stripes == 1, cap=cap)
        if world > 1:  # every rank the same round size and groups
            agreed = torch.tensor([auto_walks, auto_group], dtype=torch.int64, device="cuda")
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            auto_walks, auto_group = int(agreed[0]), int(agreed[1])
        if not args.round_walks:
            args.round_walks = stripes * auto_walks
            # equal rounds (as gn2v_train_blocks cuts an epoch): 20 steps of 2^20 walks are three
            # rounds of 6.99 M, not two of 2^23 and a half one
            total = args.steps * args.walks
            if total > args.round_walks:
                n_rounds = -(-total // args.round_walks)
                args.round_walks = -(-total // n_rounds)
        blocks.group_parts = max(1, min(args.group_parts or auto_group, blocks.parts))
        # warm-up and timed rounds share their buffers (a phantom rank is handed all ranks' walks)
        blocks.round_capacity = (min(args.round_walks, (args.warmup + args.steps) * args.walks)
                                 // stripes)

    peer_walks = {}  # phantom rank: (first walk id, walks per rank) of a timed round -> all ranks' walks

    def block_rounds(offset, total):
        """(make_walks, seed, epoch, lr, first_walk) of the rounds that train this rank's walks
        [offset, offset + total) of the run: rounds of --round-walks walks per rank (a round may
        span several steps: the longer the round, the more pairs of a centre meet in a cell);
        round ids are contiguous: [first, first + world * n) with rank r generating its n-th."""
        out, done = [], 0
        while done < total:
            nw = min(args.round_walks, total - done)
            first = (offset + done) * t_world

            def make(first=first, nw=nw):
                if phantom:  # the walks of every rank of the round, as the all-gather returns them
                    full = peer_walks.get((first, nw))
                    if full is None:  # warm-up: every rank's walks generated here
                        return ops.walks(graph, wp, 42, 0, first, t_world * nw, device=local)
                    # timed rounds: the peers' walks exist already (a fabric would deliver them);
                    # this rank generates its own share, as every rank of a real job does
                    full[t_rank * nw:(t_rank + 1) * nw].copy_(
                        ops.walks(graph, wp, 42, 0, first + t_rank * nw, nw, device=local))
                    return full
                return ops.walks(graph, wp, 42, 0, first + rank * nw, nw, device=local)

            out.append((make, 42, 0, 0.01, first))
            done += nw
        return out

    def run_steps(first_step, n_steps):
        if c_entry:
            # the whole of the reference's one call: tables initialised, alias tables built, the
            # walks of n_steps steps generated and trained round by round, tables back in node
            # order -- all of it inside the timed region
            import ctypes as C

            if n_steps == 0:
                return
            # the warm-up sizes the handle's round buffers for the timed call too (the handle keeps
            # them between fits): no memory is touched for the first time inside the timed region
            os.environ["GN2V_ROUND_BUFFERS_FOR"] = str(args.steps * args.walks)
            handle = graph.device_graph(local).handle
            stream = torch.cuda.current_stream().cuda_stream
            L = _lib.lib()
            if args.round_walks or args.stripes:
                _lib.check(L.gn2v_train_blocks(handle, C.byref(wp), C.byref(tp), 42,
                                               n_steps * args.walks, args.round_walks,
                                               args.stripes, central.data_ptr(),
                                               contextual.data_ptr(), C.byref(c_stats), stream))
            else:
                _lib.check(L.gn2v_train(handle, C.byref(wp), C.byref(tp), 42, n_steps * args.walks,
                                        central.data_ptr(), contextual.data_ptr(),
                                        C.byref(c_stats), stream))
            return
        if blocks is not None:
            blocks.run(block_rounds(first_step * args.walks, n_steps * args.walks), overlap=overlap,
                       timed=world > 1)
            return
        train = ops.cbow_step if cbow else ops.sgns_step
        for index in range(first_step, first_step + n_steps):
            # this rank's slice of the step's walk ids, trained in launches of args.batch walks
            first, count = walk_slice(index, rank, world, args.walks)
            for woff in range(0, count, args.walk_batch):
                nw = min(args.walk_batch, count - woff)
                wk = ops.walks(graph, wp, 42, 0, first + woff, nw, device=local)
                for off in range(0, nw, args.batch):
                    nb = min(args.batch, nw - off)
                    train(graph, tp, wk[off:off + nb], 42, 0, first + woff + off, 0.01, central,
                          contextual)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def memlog(tag):
        if os.environ.get("GN2V_BENCH_MEMLOG"):
            print(f"[mem] {tag}: allocated {torch.cuda.memory_allocated() / 1e9:.1f} GB, reserved "
                  f"{torch.cuda.memory_reserved() / 1e9:.1f} GB, peak allocated "
                  f"{torch.cuda.max_memory_allocated() / 1e9:.1f} GB", file=sys.stderr, flush=True)

    memlog("before warm-up")
    phase("warmup")
    fence()
    t_cold = time.perf_counter()
    run_steps(0, args.warmup)
    fence()
    cold_seconds = time.perf_counter() - t_cold
    if blocks is not None and not c_entry and getattr(blocks, "_side", None) is not None:
        # The Python trainer takes a round's walks (its own, the gathered ones, the placed ones)
        # from torch's caching allocator, on its preparation stream; a warm-up shorter than a
        # round leaves blocks of the wrong size there and the timed rounds would call hipMalloc
        # for tens of GB (1-2 s, longest on a box whose memory was never touched).  As the C
        # entry's warm-up sizes the handle's buffers for the timed call (GN2V_ROUND_BUFFERS_FOR),
        # this fills the allocator's pool with a full round's blocks before the timed region.
        full = min(args.round_walks, args.steps * args.walks)
        if full > args.warmup * args.walks:
            torch.cuda.empty_cache()
            try:
                with torch.cuda.stream(blocks._side):
                    shared = (2 if blocks.permute else 1) if (phantom or t_world > 1) else 0
                    hold = [] if phantom else [torch.empty((full, 128), dtype=torch.int32,
                                                           device="cuda")]
                    hold += [torch.empty((t_world * full, 128), dtype=torch.int32, device="cuda")
                             for _ in range(shared)]
                    for t in hold:
                        t.zero_()  # touched, not only mapped
                    del hold
            except torch.cuda.OutOfMemoryError:  # no room to spare: the timed rounds allocate as they go
                torch.cuda.empty_cache()
            fence()
    if phantom and blocks is not None:
        # A rank of a real job generates ITS walks and receives the peers' over the fabric; the
        # phantom rank used to generate all of them inside the timed region (a rank of 8: 18.8 ms
        # of walk kernel a round instead of 2.4, 5 % of its training time).  The peers' walks of
        # the timed rounds are generated here, before the clock starts.
        for make, _, _, _, first in block_rounds(args.warmup * args.walks, args.steps * args.walks):
            nw = make.__defaults__[1]
            peer_walks[(first, nw)] = ops.walks(graph, wp, 42, 0, first, t_world * nw, device=local)
        fence()
    memlog("after warm-up")
    ops.stats_reset(graph, local)
    phase("timed")
    t0 = time.perf_counter()
    if phantom:
        comm.hop_ms()  # forget the warm-up's hops
    run_steps(args.warmup, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    phase("report")
    hop_stats = comm.hop_ms() if phantom else None
    memlog("after the timed steps")
    st = ops.stats_read(graph, local)

    hop_waits = blocks.hop_wait_ms() if blocks is not None and world > 1 else []
    phases_ms = blocks.phase_ms() if blocks is not None and world > 1 else {}
    times = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    counts = torch.tensor([st["pairs"], st["walk_steps"], st["centres"]], dtype=torch.float64,
                          device="cuda")
    per_rank_pairs = [st["pairs"]]
    comm_info = None
    if world > 1:
        mine = counts[:1].clone()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank_pairs = [int(t[0]) for t in gathered]
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    if replicas:
        comm_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "per_rank_pairs": per_rank_pairs,
                     "note": "replicas only: every rank trains its own tables on its own walk ids; "
                             "nothing is exchanged inside the timed region"}
    elif world > 1:
        # the two exchanges of a round, timed alone after the measured region (they overlap with
        # training inside it): the all-gather of one round's walks and one half-partition hop
        wk = torch.zeros((min(args.round_walks, args.steps * args.walks), 128), dtype=torch.int32,
                         device="cuda")
        half = next(iter(blocks.held.values()))
        recv = torch.empty_like(half)
        reps = 4
        ms = []
        for fn in (lambda: comm.all_gather(wk),
                   lambda: comm.sendrecv_start(half, (rank - 1) % world, recv,
                                               (rank + 1) % world).wait()):
            fn()
            fence()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            fence()
            ms.append((time.perf_counter() - t1) / reps * 1e3)
        # per rank: HBM peak and what the compute stream really waited for its hops (HIP events
        # around every pending.wait(): a hop had a whole episode to complete)
        phase_names = ["walk_generation", "walk_allgather", "extract_and_sort", "training",
                       "exposed_preparation_wait"]
        mine = torch.tensor([torch.cuda.max_memory_allocated() / 1e9,
                             sum(hop_waits) / max(len(hop_waits), 1), max(hop_waits, default=0.0),
                             sum(hop_waits)] + [phases_ms.get(k, 0.0) for k in phase_names],
                            dtype=torch.float64, device="cuda")
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        comm_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "per_rank_pairs": per_rank_pairs,
                     "per_rank_hbm_peak_gb": [round(float(r[0]), 2) for r in rows],
                     "exposed_hop_wait_ms": {"hops_per_rank": len(hop_waits),
                                             "mean_per_rank": [round(float(r[1]), 4) for r in rows],
                                             "max_per_rank": [round(float(r[2]), 4) for r in rows],
                                             "total_per_rank": [round(float(r[3]), 2) for r in rows]},
                     # HIP events inside the timed region, per rank (ms, totals): the first
                     # three run on the preparation stream beside the training; `training`
                     # includes the hop waits above; `exposed_preparation_wait` is what the
                     # compute stream stood still for the preparation (walk gather included)
                     "phase_ms_per_rank": {k: [round(float(r[4 + i]), 2) for r in rows]
                                           for i, k in enumerate(phase_names)},
                     "walks_per_round_per_rank": min(args.round_walks, args.steps * args.walks),
                     "parts_per_group": blocks.group_parts,
                     "walk_allgather_ms_alone": ms[0],
                     "walk_allgather_bytes_per_rank": wk.numel() * 4,
                     "half_partition_hop_ms_alone": ms[1],
                     "half_partition_bytes": half.numel() * 4,
                     "hops_per_round": blocks.parts}
    elapsed = float(times[0])
    total_pairs, total_steps, total_centres = (float(x) for x in counts)

    if rank == 0:
        def finite(table):  # in slabs: isfinite of a 51 GB table would allocate as much again
            return all(bool(torch.isfinite(slab).all()) for slab in table.split(1 << 20))

        if c_entry:  # what the C loop planned (gn2v_stats of the timed call)
            class _Plan:
                parts, slices = c_stats.block_parts, c_stats.block_slices
                group_parts = c_stats.block_group_parts
            assert _Plan.parts, "gn2v_train did not take the block path"
            blocks_view, stripes = _Plan, max(1, c_stats.block_stripes)
            args.round_walks = c_stats.block_round_walks * stripes
        else:
            blocks_view = blocks
        if blocks is not None:
            ok = finite(blocks.central) and all(finite(t) for t in blocks.held.values())
        else:
            ok = finite(central) and finite(contextual)
        launches = max(st["train_launches"], 1)
        launch_ms = st["train_ms"] / launches
        if cbow:
            # per centre: c context rows + (1 + k) output rows, each read once and written once
            algo_bytes = 2 * 4 * d * (st["pairs"] + st["centres"] * 11)
            unit_name, value = "centres/s", total_centres / elapsed
        else:
            algo_bytes = st["pairs"] * (BYTES_PER_PAIR * d // 128)
            unit_name, value = "pairs/s", total_pairs / elapsed
        achieved = algo_bytes / (st["train_ms"] * 1e-3) / 1e9
        # What this schedule moves when nothing hits in L2: every sample row (context + k
        # negatives) read and written per pair, the central row once per RUN of equal centre (the
        # block kernel counts its runs; the walk-ordered kernels hold the centre for its window)
        row_bytes = 2 * 4 * ld
        if cbow:
            sched_bytes = algo_bytes
        elif blocks_view is not None:
            sched_bytes = row_bytes * (st["pairs"] * 11 + st["centres"])
        else:
            sched_bytes = row_bytes * (st["pairs"] * 11 + st["centres"])
        scheduled = sched_bytes / (st["train_ms"] * 1e-3) / 1e9
        if cbow:
            # mirror of launch_train's choice (gn2v_api.hip): the lazy window needs its LDS --
            # (2w + 3) rows + walk + samples + bookkeeping words per wave; workgroups of 4 / 2 / 1
            # waves within 64 KB -- and runs while a CU's 160 KB hold kLazyMinWaves = 3 waves
            lazy_bytes = 4 * ((13 * ld + 128 + 2 * 11 + 10 + 33 + 3) & ~3)
            store_mode = args.mode in ("write_through", "write_back") or (
                args.mode == "auto" and n * ld >= (1 << 22))  # GN2V_CBOW_STORES_MIN_ELEMENTS
            lazy_waves = max((160 * 1024 // ((wpb * lazy_bytes + 1023) & ~1023)) * wpb
                             for wpb in (4, 2, 1) if wpb * lazy_bytes <= 64 * 1024) \
                if lazy_bytes <= 64 * 1024 else 0
            kernel = ("gn2v::cbow_lazy_kernel" if store_mode and lazy_waves >= 3
                      else "gn2v::cbow_kernel")
        elif blocks_view is not None:
            # what the library launched (gn2v_stats.resident_launches), not a mirror of its rule
            resident_launches = (c_stats.resident_launches if c_entry
                                 else st.get("resident_launches", 0))
            # (GN2V_RESIDENT_V2=0 launches round 4's form of the resident kernel: named as such)
            resident_name = ("gn2v::sgns_resident_v2_kernel"
                             if os.environ.get("GN2V_RESIDENT_V2", "1") != "0"
                             else "gn2v::sgns_resident_kernel")
            kernel = resident_name if resident_launches else "gn2v::sgns_block_kernel"
        elif n >= (1 << 16) and args.mode in ("auto", "write_through", "write_back"):
            kernel = "gn2v::sgns_cached_kernel"
        else:
            kernel = "gn2v::sgns_kernel"
        line = {
            "metric": ("CBOW training-centres/sec (21 504 B per centre at 10 contexts), d=128"
                       if cbow else
                       "SkipGram training-pairs/sec + walk-steps/sec, d=128 (value = pairs/s with walk "
                       "generation inside the timed region; walk-steps/s in walk_kernel_steps_per_s)"),
            "value": value,
            "unit": unit_name,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"Barabasi-Albert {n} nodes / {graph.get_number_of_directed_edges() // 2}"
                            f" edges (seed 42), Node2Vec {'CBOW' if cbow else 'SkipGram'} d={d}, "
                            f"walk_length 128, window 5,"
                            f" 10 negatives, return_weight {args.return_weight}, explore_weight "
                            f"{args.explore_weight}, max_neighbours "
                            f"{args.max_neighbours or None}, {args.walks} walks per step per GPU",
                "update_mode": args.mode,
                "walks_per_launch": args.batch if blocks_view is None else None,
                # the host loop that drove the timed region
                "entry": ("gn2v_train (C ABI, include/gn2v.h: one call = tables initialised + "
                          "alias tables + all rounds + node order restored)" if c_entry else
                          "embiggen_amd.distributed.BlockPartitionedTrainer (Python; on one GPU its rounds are gn2v_block_round calls, the loop gn2v_train_blocks runs)"
                          if blocks is not None else "per-launch step entry points"),
                # what a committed PMC profile must share with this run to price its traffic
                "traffic_key": (f"ba{n}x{args.m}:d{d}:{args.model}:{args.mode}:"
                                + (f"blocks{t_world}:{blocks_view.parts}x{blocks_view.slices}:"
                                   f"round{min(args.round_walks, args.steps * args.walks)}"
                                   + (f":stripes{stripes}" if stripes > 1 else "")
                                   if blocks_view is not None else f"walk-ordered:{args.batch}")),
                "parallelism": {
                    "single": (f"{world} independent replicas (CBOW does not shard), walk-ordered "
                               "kernel" if replicas else "1 GPU, walk-ordered kernel"),
                    "blocks": f"{t_world} GPU(s), central table striped over the ranks, contextual "
                              f"table in {blocks_view.parts if blocks_view else 0} travelling parts x "
                              f"{blocks_view.slices if blocks_view else 0} "
                              + ("resident cells (every contextual row in the LDS of the one "
                                 "workgroup that owns its cell)"
                                 if blocks_view is not None and blocks_view.slices > 16
                                 else "XCD slice(s) (no shared rows)") + ", rounds of "
                              f"{min(args.round_walks, args.steps * args.walks)} walks per GPU "
                              f"prepared {blocks_view.group_parts if blocks_view else 0} parts at a time"
                              + (f" trained in {stripes} centre stripes" if stripes > 1 else "")
                              + ((", a round = one extraction group = one launch of all its cells "
                                  "(the next round is prepared on a second lane of round buffers "
                                  "only while that second set stays under 16 GiB: "
                                  "gn2v_train_blocks)"
                                  if blocks_view is not None and stripes <= 1
                                  and blocks_view.group_parts >= blocks_view.parts
                                  else ", the next group prepared beside the training (second pair "
                                       "buffer, own stream)")
                                 if c_entry and os.environ.get("GN2V_BLOCK_OVERLAP", "1") != "0"
                                 else f", preparation "
                                      f"{'overlapped' if overlap and stripes == 1 else 'in line'}"),
                }[mode],
            },
            "pairs_per_s": total_pairs / elapsed,
            "walk_steps_per_s": total_steps / elapsed,
            "walk_kernel_steps_per_s": st["walk_steps"] / max(st["walk_ms"] * 1e-3, 1e-12),
            "finite": ok,
            "argv": sys.argv[1:],
            "roofline": {
                "bound": "hbm",
                "kernel": kernel,
                # bandwidth as bandwidth (filled below from the committed PMC passes): bytes that
                # really leave L2 per second; a fraction of the 8 TB/s peak, never above 1
                "achieved_hbm": None,
                "frac_hbm": None,
                "traffic": None,
                # SURVEY 8d's schedule-independent model (12 288 B per pair: every row of every
                # pair read and written in HBM).  `frac` can exceed what the memory system moves
                # -- even 1 -- when the schedule keeps the centre in registers and the XCD's L2
                # serves hub rows: it prices work, `frac_hbm` prices bandwidth
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # the same with the central row counted once per run of equal centre
                "scheduled": scheduled,
                "frac_scheduled": scheduled / HBM_PEAK_GBS,
                "mean_centre_run": st["pairs"] / max(st["centres"], 1),
                "algorithmic_bytes_per_launch": algo_bytes // launches,
                "scheduled_bytes_per_launch": sched_bytes // launches,
                "avg_launch_ms": launch_ms,
                "launches": st["train_launches"],
                "note": "frac_hbm = bytes that leave L2 (rocprofv3 PMC, committed profile of this "
                        "workload) / time / peak: bandwidth, never above 1.  frac = SURVEY 8d's "
                        "schedule-independent bytes (every row of every pair read and written in "
                        "HBM) / time / peak: it prices work and may exceed 1 because the centre row "
                        "of a run stays in registers and the XCD's L2 serves hub rows -- and, with "
                        "resident cells (gn2v::sgns_resident_kernel), because every contextual row "
                        "of a launch lives in the LDS of the workgroup that owns its cell: only the "
                        "central rows and the pair words move through HBM, and the kernel is bound "
                        "by LDS latency and instruction issue, not by HBM; "
                        "frac_scheduled counts the centre once per run",
            },
        }
        if kernel == "gn2v::sgns_resident_v2_kernel":
            # Resident cells: the contextual rows of a launch live in LDS, HBM carries the central
            # rows and the pair words only -- the byte model of SURVEY 8d does not bound this
            # kernel (its "fraction" came out at 3).  What does: every pair hands its
            # gradient to its central row with `ld` f32 atomic adds, and the L2 atomic units
            # retire 3.3e11 of them per second (round 5: without the hand-over the same kernel
            # ran at 3.5e9 pairs/s, with it at 2.2e9; the kernel of the round's end 2.5e9 against
            # 2.3e9: profiles/r05_logs/r5_cold_centre_stores.log).  frac = atomic dwords
            # per second / that ceiling.  The byte figures stay beside it under explicit names,
            # and so does the share of the vector pipes' issue rate the arithmetic needs.
            roof = line["roofline"]
            floor = valu_floor_per_pair(ld, 10)
            kernel_pairs_per_s = st["pairs"] / (st["train_ms"] * 1e-3)
            roof["work_over_hbm_peak"] = roof["frac"]
            roof["work_bytes_per_s_gb"] = roof["achieved"]
            roof["bound"] = "l2_atomic"
            roof["unit"] = "G f32 atomic adds/s"
            roof["peak"] = L2_ATOMIC_PEAK_GDWORDS
            roof["achieved"] = ld * kernel_pairs_per_s / 1e9
            roof["frac"] = roof["achieved"] / L2_ATOMIC_PEAK_GDWORDS
            roof["atomic_dwords_per_pair"] = ld
            roof["kernel_pairs_per_s"] = kernel_pairs_per_s
            roof["valu_floor_per_pair"] = floor
            roof["valu_issue_frac"] = floor * kernel_pairs_per_s / 1e9 / VALU_ISSUE_PEAK_GIPS
            # counters behind the two figures above, from the committed passes of this kernel
            # (scripts/atomic_counters.sh, scripts/resident_counters.sh): how busy the L2 atomic
            # units really are, and the vector instructions really issued per pair
            roof.update(committed_counters(kernel, ld, kernel_pairs_per_s))
            roof["note"] = (
                "bound = l2_atomic: frac = (ld f32 atomic adds per pair: the gradient of the "
                "central row, exact) x kernel pairs/s / 3.31e11 dword adds/s (what "
                "scripts/atomic_probe.hip measures on this chip for any instruction shape; "
                "profiles/r05_logs/r5_atomic_probe.log, counters profiles/r06_atomic_summary.md).  valu_issue_frac = the arithmetic's "
                f"floor of vector instructions per pair ({floor:.1f}: bench.py "
                "valu_floor_per_pair) x kernel pairs/s / (1 024 SIMDs x 2.4 GHz / 4 cycles): the "
                "counter pass profiles/r06_resident_counters.json gives the instructions really "
                "issued (valu_issued_frac); atomic_unit_busy = the TCC counters of this kernel "
                "next to the probe's (profiles/r06_atomic_summary.md).  work_over_hbm_peak = SURVEY 8d's 12 288 B per pair / time / 8 TB/s: it "
                "prices work, exceeds 1 because the sample rows never move through HBM, and is "
                "not a roofline fraction; frac_hbm = bytes that really leave L2 (committed PMC "
                "profile) / time / 8 TB/s")
        # What `value` leaves out: the FIRST fit on a fresh handle also builds the walk sampler's
        # edge records and allocates the round buffers (tens of GB the driver has to clear); the
        # warm-up is that call here -- and it sizes the buffers for the timed call
        # (GN2V_ROUND_BUFFERS_FOR), which therefore touches no memory for the first time.
        # first_fit_s = the warm-up call, cold; overhead_s = that minus its steps at the timed rate.
        line["first_fit_s"] = cold_seconds
        line["first_fit"] = {
            "seconds": cold_seconds, "steps": args.warmup,
            "overhead_s": cold_seconds - args.warmup * elapsed / max(args.steps, 1),
            "note": "the warm-up call on a fresh graph handle (sampler set-up + round-buffer "
                    "allocation included); `value` is the steady state of a handle's later fits"}
        if reserved is not None:
            line["config"]["reserved_cus_per_xcd"] = args.reserve_cus
            line["config"]["active_cus_per_xcd"] = reserved
        if comm_info is not None:
            line["distributed"] = comm_info
        line["hbm_peak_gb"] = {"torch_allocated": torch.cuda.max_memory_allocated() / 1e9,
                               "torch_reserved": torch.cuda.max_memory_reserved() / 1e9,
                               "device_in_use_now": (lambda f, t: (t - f) / 1e9)(*torch.cuda.mem_get_info())}
        if phantom:
            line["metric"] = "PHANTOM RANK (not the benchmark value): " + line["metric"]
            line["phantom"] = {
                "world": t_world, "rank": t_rank,
                "pairs_this_rank": st["pairs"],
                "walk_steps_generated_in_the_timed_region": st["walk_steps"],
                "hop_copies": {"count": hop_stats[0], "mean_ms": hop_stats[1], "max_ms": hop_stats[2],
                               "bytes_each": int(next(iter(blocks.held.values())).numel() * 4),
                               "what": "a part-sized device copy on its own stream, issued like "
                                       "an RCCL hop while the training kernel holds the CUs"},
                "note": "one rank of that world on one GPU, no fabric: value = this rank's pairs/s; "
                        "this rank's walks are generated inside the timed region, the peers' before "
                        "it (a fabric would deliver them; no all-gather is timed here)"}
        pmc, why_not = committed_traffic(line["config"]["traffic_key"])
        if pmc is not None:
            ratio, source = pmc
            bytes_per_launch = ratio * (algo_bytes / launches)
            traffic = bytes_per_launch / (launch_ms * 1e-3) / 1e9
            line["roofline"]["traffic"] = traffic
            line["roofline"]["achieved_hbm"] = traffic
            line["roofline"]["frac_hbm"] = traffic / HBM_PEAK_GBS
            line["roofline"]["traffic_over_algorithmic"] = ratio
            line["roofline"]["traffic_bytes_per_launch"] = bytes_per_launch
            line["roofline"]["traffic_source"] = f"profiles/{source} (rocprofv3 --pmc FETCH_SIZE / " \
                                                 "WRITE_SIZE, calibrated; same workload and schedule, " \
                                                 "rescaled by this run's launch time)"
        else:
            line["roofline"]["traffic_missing_reason"] = why_not
        if world == 1 and not args.no_cpu_baseline and not phantom:
            if blocks is not None:
                central, contextual = blocks.gather_full()
            line["cpu_baseline"] = cpu_baseline(graph, args, central, contextual, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL logs its banner through C stdio on stdout; flush it so the JSON line comes last
    import ctypes

    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
