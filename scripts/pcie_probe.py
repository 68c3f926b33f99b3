import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import embiggen_amd as E
from embiggen_amd import ops
g = E.barabasi_albert(10_000_000, 10, 42)
rp, ci = g.row_ptr, g.col_idx  # host copies
t0 = time.perf_counter(); h = E.CSRGraph.from_csr(rp, ci); h.device_graph(0); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"CSR upload {(rp.nbytes + ci.nbytes) / 1e9:.2f} GB in {t1 - t0:.3f} s")
c = ops.init_table(10_000_000, 128, 42, 0, 0.1); x = ops.init_table(10_000_000, 128, 42, 1, 0.1); torch.cuda.synchronize()
t0 = time.perf_counter(); a = c.cpu().numpy(); b = x.cpu().numpy(); t1 = time.perf_counter()
print(f"tables download {(a.nbytes + b.nbytes) / 1e9:.2f} GB in {t1 - t0:.3f} s")
