#!/usr/bin/env python3
"""BASELINE configs[3] on N GPUs: the index of the 2^28-byte text is built on rank 0,
replicated by one RCCL broadcast, the 10^7 patterns are cut into N contiguous slices, counts gathered
in pattern order.  `python scripts/fm_multi_bench.py --gpus N` starts its own ranks (as bench.py does);
TC_BENCH_REHEARSAL=1: all ranks on cuda:0 over gloo (control flow only).  Rank 0 prints one JSON line."""
import argparse, ctypes as C, json, os, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--text-bytes", type=int, default=1 << 28)
ap.add_argument("--patterns", type=int, default=10_000_000)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
if a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # parent: never touches the GPU, never execs
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "4")
    sys.exit(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                             "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
                             "--gpus", str(a.gpus), "--text-bytes", str(a.text_bytes), "--patterns", str(a.patterns),
                             "--iters", str(a.iters)], env=env).returncode)
import torch, torch.distributed as dist, textcomp
from textcomp.fmshard import replicate_index, sharded_count
from textcomp.synth import c4_patterns_dev
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
rehearsal = os.environ.get("TC_BENCH_REHEARSAL") == "1"
local = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group("gloo") if rehearsal else dist.init_process_group("nccl", device_id=torch.device("cuda", local))
ctx = textcomp.Context(local)
n, npat = a.text_bytes, a.patterns
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
assert ctx.lib.tc_generate_dev(ctx.handle, 0, 0xC4, n, C.c_void_p(d_text.data_ptr())) == 0
torch.cuda.synchronize()
pats, d_offs = c4_patterns_dev(ctx, d_text, npat)
flat = pats.reshape(-1)
t0 = time.perf_counter()
fm = ctx.fm_build(d_text.cpu().numpy()) if rank == 0 else None
t_build = time.perf_counter() - t0
t0 = time.perf_counter()
mine = replicate_index(ctx, fm, src=0) if world > 1 else fm
torch.cuda.synchronize(); t_repl = time.perf_counter() - t0
best = 1e9
for it in range(a.iters + 1):
    if world > 1: dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = sharded_count(ctx, mine, flat, d_offs, npat) if world > 1 else mine.count_dev(flat, d_offs, npat)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    if it: best = min(best, dt)
if rank == 0:
    hits = int((out > 0).sum().item())
    assert hits == npat - npat // 100, hits
    print(json.dumps({"metric": "FM-index count, patterns/s", "value": round(npat / best / 1e6, 1), "unit": "Mpatterns/s",
                      "n_gpus": world, "ms_per_batch": round(best * 1e3, 3), "text_bytes": n, "patterns": npat,
                      "index_build_ms": round(t_build * 1e3, 1), "index_broadcast_ms": round(t_repl * 1e3, 1),
                      "found": hits, "scaling": "strong", "backend": dist.get_backend() if world > 1 else None}), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
