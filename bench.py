#!/usr/bin/env python3
"""bench.py -- BWT+MTF+RLE encode throughput on synthetic ACGTN records.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the command launches itself: the parent (which never
touches the GPU) starts N fresh ranks through torch.distributed.run on 127.0.0.1, forwards rank 0's
JSON line and returns the children's exit code.  Under torchrun (WORLD_SIZE set) it is a rank.

A "step" = one fused BWT -> MTF -> RLE encode of this rank's record, input and output resident in
HBM.  N = 1: tc_encode_dev (the runs as arrays: BASELINE configs[2]).  N > 1: each rank owns one
independent record (seed 0xC500 + rank) and encodes it straight into its container
(tc_encode_container_dev: the RLE stage writes the wire format, nothing is packed afterwards), and the step
ends with the gather of the containers on rank 0 over RCCL, overlapped with the next record's encode.
Rank 0 prints ONE JSON line.  After the timed region, at N = 1, the line also gets `container` (the N > 1
step without its exchange, timed on this one GPU) and `fm_count` (BASELINE configs[3]).

PyTorch here is plumbing only: device buffers, torch.distributed (nccl = RCCL),
barriers.  The compute is libtextcomp.so through its C ABI.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))

GIB = 1 << 30


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--record-bytes", dest="n", type=int, default=GIB, help="record size in bytes (default 1 GiB = BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fm", action="store_true", help="skip the FM-index count leg (BASELINE configs[3])")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-pointer leg (tc_encode_container, pageable and page-locked buffers)")
    ap.add_argument("--no-classes", action="store_true", help="skip the input-classes leg (non-iid records at 2^28 bytes)")
    ap.add_argument("--cpu-sample", type=int, default=64 << 20, help="bytes of the workload timed on the CPU port")
    return ap.parse_args()


def algorithmic_bytes(st, n):
    """A = A_sa + A_bwt + A_mtf + A_rle from the counters the library reports
    (R, m_r, k_r, P_r); DESIGN.md section 5 states the per-kernel terms."""
    N = n + 1
    a_sa = n
    for r in range(st.rounds):
        m, k, P = int(st.m[r]), int(st.key_bytes[r]), int(st.passes[r])
        a_sa += m * ((k + 4) * 2 * P + k) + m * (k + 12) + m * (k + 8)
    return a_sa + 6 * N + 2 * N + N + 5 * int(st.runs)


def radix_launch_bytes(st, n):
    """Algorithmic bytes of the timed round-0 radix-pass launches: (k+4)*2 = 24 B per suffix
    per pass, except a fused first pass, which reads the text (1 B) instead of a key array
    (12 B).  Returns the mean per launch."""
    N = n + 1
    L = int(st.radix_launches)
    if L == 0:
        return 0
    if getattr(st, "msd_keyonly", 0):      # keys only: 8 B read + 8 B written per suffix; the first level reads the text
        return (16 * N * L - 7 * N) / L
    total = 24 * N * L
    if st.keygen_fused:
        total -= 11 * N
    return total / L


def haskell_probe():
    """north_star asks for the reference's Haskell path timed beside the GPU number (built per
    text-compression.cabal:98).  That needs GHC on this host; look for it on PATH only (no exec, no
    subprocess: this also runs in processes that have initialised the GPU)."""
    import shutil
    found = {t: shutil.which(t) for t in ("ghc", "cabal", "stack", "runghc")}
    if found["ghc"]:
        return "ghc found at %s but the reference sources are not on this host (nproc=%d): not measured" % (found["ghc"], os.cpu_count())
    return "not measurable: ghc/cabal/stack absent on this host (nproc=%d); the reference sources do not travel either" % os.cpu_count()


def self_launch(a):
    """bare `python bench.py --gpus N` (the driver's form): start N ranks as CHILD processes.  Nothing
    in this process has touched the GPU (no torch import yet), and it never execs.

    The first N > 1 exchange over RCCL cannot be rehearsed on the builder's one-GPU boxes, so the parent makes sure
    the driver gets a line whatever happens to it: the ranks run in their own process group under a DEADLINE
    (300 s start-up allowance -- a fresh box pages torch in for a minute or two, N ranks at once -- + (warm-up + steps) x 10 x the N = 1 step of a record of this size + the verification of the
    gathered containers; TC_BENCH_DEADLINE_S overrides).  If they exceed it, or exit non-zero, they are killed as a
    group and FRESH ranks are started once with the most conservative exchange (TC_BENCH_GATHER=torch: the
    torch.distributed batch on torch's stream; TC_COMM_CUS=0: no CU-restricted stream); the line then says so
    (`gather.path`, `gather.fallback_reason`).  If that attempt fails as well the parent exits non-zero with the
    tails of both."""
    import signal
    import socket
    import subprocess
    import tempfile

    def attempt(extra_env, deadline):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        env["TC_BENCH_SELF_LAUNCHED"] = "1"
        env.update(extra_env)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
               # (rebuilt from the parsed values: torchrun's own parser chokes on abbreviations such as --n)
               "--gpus", str(a.gpus), "--steps", str(a.steps), "--warmup", str(a.warmup), "--record-bytes", str(a.n),
               "--cpu-sample", str(a.cpu_sample)] + (["--no-cpu-baseline"] if a.no_cpu_baseline else []) + (["--no-fm"] if a.no_fm else []) + (["--no-classes"] if a.no_classes else []) + (["--no-host-path"] if a.no_host_path else [])
        if os.environ.get("TC_BENCH_CHILD_CMD"):      # test hook (CPU suite): the parent's deadline / fallback logic around any child
            import shlex
            cmd = shlex.split(os.environ["TC_BENCH_CHILD_CMD"])
        with tempfile.TemporaryFile(mode="w+") as fo, tempfile.TemporaryFile(mode="w+") as fe:
            p = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, start_new_session=True)
            why = None
            try:
                rc = p.wait(timeout=deadline)
                if rc != 0:
                    why = "exit code %d" % rc
            except subprocess.TimeoutExpired:
                why = "no result within the deadline of %.0f s" % deadline
                rc = -1
            if why is not None:        # the whole group: torchrun and every rank it started
                for sig in (signal.SIGTERM, signal.SIGKILL):
                    try:
                        os.killpg(p.pid, sig)
                    except (ProcessLookupError, PermissionError):
                        break
                    try:
                        p.wait(timeout=10)
                        break
                    except subprocess.TimeoutExpired:
                        continue
            fo.seek(0)
            fe.seek(0)
            return why, fo.read(), fe.read()[-6000:]

    step_est = max(0.005, 0.025 * a.n / GIB)          # the N = 1 step of a record of this size (25 ms per GiB), seconds
    deadline = float(os.environ.get("TC_BENCH_DEADLINE_S", "0")) or (300.0 + (a.warmup + a.steps) * 10.0 * step_est + 2.0 * a.gpus)
    why, out, err = attempt({}, deadline)
    if why is not None:
        sys.stderr.write("bench.py: the ranks failed (%s); last lines:\n%s\nbench.py: starting fresh ranks with TC_BENCH_GATHER=torch TC_COMM_CUS=0\n" % (why, err[-3000:]))
        why2, out, err2 = attempt({"TC_BENCH_GATHER": "torch", "TC_COMM_CUS": "0", "TC_BENCH_FALLBACK_REASON": why}, deadline)
        if why2 is not None:
            sys.stderr.write("bench.py: the fallback ranks failed as well (%s); last lines:\n%s\n" % (why2, err2[-3000:]))
            sys.exit(1)
    sys.stdout.write(out)
    sys.stdout.flush()
    sys.exit(0)


def cpu_baseline(n_sample, seed):
    """The CPU restatement (oracle/, kind "port", 1 thread) on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    t = O.gen_acgtn(seed, n_sample)
    t0 = time.perf_counter()
    L = O.bwt_encode_arr(t)
    idx, _ = O.mtf_encode_arr(L)
    O.rle_encode_u32_arr(idx)
    dt = time.perf_counter() - t0
    return {"value": round(n_sample / dt / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
            "reference_haskell": haskell_probe(),
            "sample": "first %d MiB of the rank-0 record, BWT+MTF+RLE encode by oracle/tc_oracle.c, %.1f s"
                      % (n_sample >> 20, dt)}


def container_leg(ctx, lib, torch, d_text, n, steps, ms_plain):
    """The step of the N > 1 path without its exchange, on this one GPU, after the timed region: the record
    encoded straight into its container (tc_encode_container_dev) -- what every rank does per step before it
    posts the gather.  `ratio` compares it with the N = 1 step (tc_encode_dev, run arrays out)."""
    pcap = n + n // 4 + 4096
    buf = torch.empty(pcap, dtype=torch.uint8, device=d_text.device)
    nb = C.c_uint64(pcap)

    def one():
        nb.value = pcap
        rc = lib.tc_encode_container_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.c_void_p(buf.data_ptr()), C.byref(nb))
        if rc != 0:
            raise RuntimeError("tc_encode_container_dev rc=%d: %s" % (rc, lib.tc_last_error(ctx.handle).decode()))
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    st = ctx.stats()
    return {"ms_per_step_with_container": round(ms, 3), "ratio_to_ms_per_step": round(ms / ms_plain, 4),
            "container_bytes": int(nb.value), "bytes_per_input_byte": round(nb.value / n, 4),
            "stages_ms": {"suffix_sort+bwt": round(st.ms_sa, 3), "mtf": round(st.ms_mtf, 3), "rle+wire_format+seal": round(st.ms_rle, 3)},
            "what": "tc_encode_container_dev: BWT -> MTF -> RLE written as the container's nibble stream by the RLE stage, "
                    "sealed on the device; bit-identical to tc_encode_dev + tc_block_to_container_dev (tests/test_gpu_container_fused.py)"}


def host_path_leg(ctx, lib, torch, d_text, n, ms_step):
    """SURVEY.md 8(d), secondary metric -- the path a Haskell caller takes (bytestringToBWT and friends hand over a host
    ByteString: BWT.hs:68-70, RLE.hs:83-85): tc_encode_container with HOST buffers in and out, the 1 GiB record, after
    the timed region.  Twice: pageable buffers (numpy: staged through the context's page-locked ring by its helper
    threads) and page-locked buffers (one asynchronous copy each way).  Best of 3 calls each; the container is compared
    byte for byte with the device path's.  `pcie_floor_ms` = this box's measured page-locked copy times of the same
    bytes + the device step: what a path with no overlap at all between copies and compute cannot beat."""
    import numpy as np
    cap = n + n // 4 + 4096
    lib.tc_encode_container.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    dref = torch.empty(cap, dtype=torch.uint8, device=d_text.device)
    nb = C.c_uint64(cap)
    assert lib.tc_encode_container_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.c_void_p(dref.data_ptr()), C.byref(nb)) == 0
    ref = dref[:nb.value].cpu()
    res = {"record_bytes": n, "container_bytes": int(nb.value), "call": "tc_encode_container (host pointers in and out)"}
    # this box's PCIe rates, page-locked, same sizes
    pin_t = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    pin_o = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
    pin_t.copy_(d_text)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); d_text.copy_(pin_t, non_blocking=True); torch.cuda.synchronize(); h2d = time.perf_counter() - t0
    t0 = time.perf_counter(); pin_o[:nb.value].copy_(dref[:nb.value], non_blocking=True); torch.cuda.synchronize(); d2h = time.perf_counter() - t0
    res["pcie_h2d_GBps"] = round(n / h2d / 1e9, 1)
    res["pcie_d2h_GBps"] = round(nb.value / d2h / 1e9, 1)
    res["pcie_floor_ms"] = round((h2d + d2h) * 1e3 + ms_step, 2)
    for kind in ("pageable", "page_locked"):
        if kind == "pageable":
            text = np.empty(n, dtype=np.uint8)
            text[:] = pin_t.numpy()
            out = np.empty(cap, dtype=np.uint8)
            tp, op = text.ctypes.data, out.ctypes.data
        else:
            text, out = pin_t, pin_o
            tp, op = pin_t.data_ptr(), pin_o.data_ptr()
        ts = []
        for _ in range(4):
            used = C.c_uint64(cap)
            t0 = time.perf_counter()
            rc = lib.tc_encode_container(ctx.handle, C.c_void_p(tp), n, C.c_void_p(op), C.byref(used))
            ts.append(time.perf_counter() - t0)
            assert rc == 0, "tc_encode_container rc=%d: %s" % (rc, lib.tc_last_error(ctx.handle).decode())
        got = torch.from_numpy(out[:used.value]) if kind == "pageable" else out[:used.value]
        best = min(ts[1:])
        res[kind] = {"ms": round(best * 1e3, 2), "MBps_host_to_host": round(n / best / 1e6, 1), "first_call_ms": round(ts[0] * 1e3, 2),
                     "identical_to_device_path": bool(used.value == nb.value and torch.equal(got, ref))}
        del text, out
    return res


CLASSES = (("genome_like", 2, 0x6E0E), ("zipf_words", 3, 0x21BF), ("runs_p0.9", 4, 0x9A75), ("repeat_4KiB", 5, 0x4B1B),
           ("acgt_gaps", 6, 0x6A95))


def classes_leg(ctx, lib, torch, n=1 << 28):
    """Away from iid ACGTN, after the timed region: five records of 2^28 bytes from the device-side generators
    (tc_generate_dev kinds 2 .. 6: repeat-rich DNA, Zipf-distributed words, runs, a 4 KiB block repeated, an assembly
    with gaps -- iid ACGT with one run of n / 64 'N's and sixteen of n / 4096), each
    encoded twice by the same call as the headline (tc_encode_dev; best of the two), decoded, and compared with the
    text on the device.  `rounds` / `m` are the prefix-doubling rounds and their tied sets (tc_stats); `chain_rounds`: how many
    of them ran as chain rounds (tc_chain.hpp: the periodic record is done in 3 rounds instead of 25)."""
    from textcomp import Block
    cap = n + 2
    d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
    d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
    res = {"record_bytes": n}
    for name, kind, seed in CLASSES:
        try:
            rc = lib.tc_generate_dev(ctx.handle, kind, seed, n, C.c_void_p(d_text.data_ptr()))
            assert rc == 0, "generate rc=%d" % rc
            torch.cuda.synchronize()
            enc, dec = [], []
            for _ in range(2):
                blk = Block()
                blk.nruns = cap
                blk.run_count = d_cnt.data_ptr()
                blk.run_value = d_val.data_ptr()
                t0 = time.perf_counter()
                rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk))
                enc.append(time.perf_counter() - t0)
                assert rc == 0, "encode rc=%d: %s" % (rc, lib.tc_last_error(ctx.handle).decode())
                st = ctx.stats()
                t0 = time.perf_counter()
                rc = lib.tc_decode_dev(ctx.handle, C.byref(blk), C.c_void_p(d_out.data_ptr()))
                dec.append(time.perf_counter() - t0)
                assert rc == 0, "decode rc=%d: %s" % (rc, lib.tc_last_error(ctx.handle).decode())
            res[name] = {"kind": kind, "seed": seed, "encode_ms": round(min(enc) * 1e3, 2), "encode_MBps": round(n / min(enc) / 1e6, 1),
                         "decode_ms": round(min(dec) * 1e3, 2), "rounds": int(st.rounds), "chain_rounds": int(st.chain_rounds),
                         "m": [int(st.m[i]) for i in range(min(int(st.rounds), 6))], "sigma": int(blk.sigma), "runs": int(blk.nruns),
                         "round_trip_exact": bool(torch.equal(d_out, d_text))}
        except Exception as e:   # noqa: BLE001
            res[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def fm_count_leg(ctx, lib, torch, no_cpu):
    """BASELINE configs[3], after the timed encode region: 10^7 x 100-byte ACGTN patterns (99 % substrings of the
    text, every 100th iid: SURVEY.md 8d) counted against the index of a 2^28-byte text, patterns and index resident
    in HBM.  The batch is one tc_fm_count_dev call; best and mean of 5 calls after a warm-up.  `steps_executed` is
    measured (a miss stops when its range empties), A_cnt = steps x 2 x 64 B + pattern bytes.  The
    same batch against a 2^29-byte text shows the rate once the index no longer fits the 256 MB Infinity Cache."""
    import numpy as np
    from textcomp.synth import c4_patterns_dev
    npat, m = 10_000_000, 100
    lib.tc_fm_count_dev.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
    res = {}
    for lg in (28, 29):
        n = 1 << lg
        d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert lib.tc_generate_dev(ctx.handle, 0, 0xC4, n, C.c_void_p(d_text.data_ptr())) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fm = ctx.fm_build_dev(d_text)      # the text is in HBM already: no round trip through the host (tc_fm_build_dev)
        t_build = time.perf_counter() - t0
        t_build_host = None
        if lg == 28:                       # once, for comparison: the host entry point (pageable buffer in, PCIe included)
            text = d_text.cpu().numpy()
            t0 = time.perf_counter()
            fmh = ctx.fm_build(text)
            t_build_host = time.perf_counter() - t0
            fmh.close()
            del text
        pats, d_offs = c4_patterns_dev(ctx, d_text, npat, m)
        d_out = torch.zeros(npat, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ts = []
        for it in range(6):
            t0 = time.perf_counter()
            rc = lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(pats.data_ptr()), C.c_void_p(d_offs.data_ptr()), npat, C.c_void_p(d_out.data_ptr()))
            dt = time.perf_counter() - t0
            assert rc == 0, lib.tc_last_error(ctx.handle)
            if it:
                ts.append(dt)
        out = d_out.cpu().numpy()
        is_miss = (np.arange(npat) % 100) == 99
        ok = bool((out[~is_miss] >= 1).all() and (out[is_miss] == 0).all())
        if lg == 29:
            res["outside_mall"] = {"text_bytes": n, "ms": round(min(ts) * 1e3, 3), "ms_mean": round(sum(ts) / len(ts) * 1e3, 3),
                                   "Mpatterns_per_s": round(npat / min(ts) / 1e6, 1), "hits_and_misses_as_generated": ok}
            fm.close()
            break
        # executed steps of the miss patterns: the shortest suffix with count 0 is the step that empties the range
        sub = np.arange(99, npat, 100)[:2000]
        mp = pats[torch.from_numpy(sub).cuda()]
        alive = np.ones(len(sub), bool)
        last = np.zeros(len(sub), np.int64)
        for ln in range(1, 48):
            sfx = mp[:, m - ln:].contiguous()
            so = (torch.arange(len(sub) + 1, device="cuda", dtype=torch.int64) * ln).contiguous()
            o = torch.zeros(len(sub), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            assert lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(sfx.data_ptr()), C.c_void_p(so.data_ptr()), len(sub), C.c_void_p(o.data_ptr())) == 0
            z = o.cpu().numpy() == 0
            last[alive & z] = ln
            alive &= ~z
            if not alive.any():
                break
        steps = float(int((~is_miss).sum()) * m + last.mean() * int(is_miss.sum()))
        A = steps * 128 + npat * m
        best = min(ts)
        res.update({"ms": round(best * 1e3, 3), "ms_mean": round(sum(ts) / len(ts) * 1e3, 3), "calls_timed": len(ts),
                    "Mpatterns_per_s": round(npat / best / 1e6, 1), "patterns": npat, "pattern_bytes": m, "text_bytes": n,
                    "index_build_ms_dev": round(t_build * 1e3, 1),
                    "index_build_ms_host_text_in": round(t_build_host * 1e3, 1) if t_build_host else None,
                    "steps_executed": int(steps), "miss_patterns_stop_after_steps": round(float(last.mean()), 2),
                    # A_cnt: the SURVEY 8(d) single-step formula (steps x 2 x 64 B + pattern bytes), kept for comparability;
                    # the pair vectors fetch about a third of it (fetch_bytes below: recorded counters)
                    "A_cnt": int(A), "A_cnt_formula_GBps": round(A / best / 1e9, 1),
                    "bound": "dependent random 64-B line reads (one lane = one pattern = a chain of lookups); A_cnt / t prices the formula, it is not an HBM figure",
                    "hits_and_misses_as_generated": ok})
        pj = os.path.join(ROOT, "profiles", "traffic_latest.json")
        try:
            tj = json.load(open(pj))
            res["fetch_bytes"] = tj.get("fm_count_kernel_fetch_bytes_per_launch")
            if res["fetch_bytes"]:
                res["fetch_GBps"] = round(res["fetch_bytes"] / best / 1e9, 1)
                res["fetch_Glines_per_s"] = round(res["fetch_bytes"] / 64 / best / 1e9, 1)
            res["fetch_bytes_source"] = "recorded: profiles/traffic_latest.json (%s), not measured in this run" % tj.get("fm_commit", tj.get("commit", "?"))
        except Exception:
            res["fetch_bytes"] = None
        fm.close()
        del d_text, pats, d_offs, d_out
    if not no_cpu:
        # the CPU port on the same workload SHAPE over a 2^24-byte text (the oracle's index of 2^28 bytes is minutes)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle as O
        from textcomp.synth import c4_offsets
        n2, np2 = 1 << 24, 200_000
        t = O.gen_acgtn(0xC4, n2)
        t0 = time.perf_counter()
        ofm = O.FMIndex(t)
        tb = time.perf_counter() - t0
        offs = c4_offsets(n2, np2, m)
        flat = t[(offs[:, None] + np.arange(m)[None, :]).reshape(-1)].reshape(np2, m).copy()
        flat[99::100] = O.gen_acgtn(0xC4F1, (np2 // 100) * m).reshape(-1, m)
        po = (np.arange(np2 + 1, dtype=np.int64) * m)
        cores = min(os.cpu_count() or 1, 64)
        t0 = time.perf_counter()
        want = ofm.count_batch(flat.reshape(-1), po, threads=cores)
        tc = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(np2 / tc / 1e6, 4), "unit": "Mpatterns/s", "cores": cores, "kind": "port",
                               "sample": "%d patterns of the same generator against the oracle's index of a 2^24-byte text "
                                         "(index build %.1f s on 1 core, count %.2f s on %d threads); found %d"
                                         % (np2, tb, tc, cores, int((want > 0).sum()))}
    return res


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    import torch
    import torch.distributed as dist
    import textcomp
    from textcomp import Block

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit("bench.py --gpus %d started with WORLD_SIZE=%d" % (a.gpus, world))
    # TC_BENCH_REHEARSAL=1: every rank computes on cuda:0 and the exchange runs over gloo with
    # host staging -- a one-GPU rehearsal of the N > 1 control flow (never a measurement)
    rehearsal = os.environ.get("TC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    xdev = torch.device("cpu") if rehearsal else dev   # where the exchanged payload lives
    if os.environ.get("TC_BENCH_FAIL_RANK") == str(rank):   # test hook: a rank that dies (exit-code propagation)
        sys.exit(3)
    # test hook: a rank that never reaches its first exchange (the parent's deadline).  TC_BENCH_STALL_RANK stalls the
    # first attempt only (the fallback ranks run through), TC_BENCH_STALL_ALWAYS every attempt.
    stall = (os.environ.get("TC_BENCH_STALL_ALWAYS") == str(rank) or
             (os.environ.get("TC_BENCH_STALL_RANK") == str(rank) and "TC_BENCH_FALLBACK_REASON" not in os.environ))

    n = a.n
    cap = n + 2
    ctx = textcomp.Context(local)
    lib = ctx.lib
    lib.tc_ctx_set_profile(ctx.handle, 1)
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    d_cnt = torch.empty(cap, dtype=torch.int32, device=dev)
    d_val = torch.empty(cap, dtype=torch.int16, device=dev)
    seed = 0xC3 if world == 1 else 0xC500 + rank
    rc = lib.tc_generate_dev(ctx.handle, 0, seed, n, C.c_void_p(d_text.data_ptr()))
    assert rc == 0, rc
    torch.cuda.synchronize()

    # N > 1: the encoded block becomes a container (header + nibble stream, ~0.42 bytes per input byte), gathered on
    # rank 0, pipelined so that the transfer of record k overlaps the encode of record k+1
    gatherer = None
    packed = None
    if world > 1:
        from textcomp.gather import BlockGather, NativeGather
        pcap = n + n // 4 + 4096          # packed bytes per record (iid ACGTN: ~0.8 n), with slack
        # the exchange goes through the library's own RCCL communicator (tc_comm_*, the C ABI a Haskell or C caller
        # would use): one group of point-to-point transfers into rank 0, posted behind the encode
        # and on its own CU-restricted stream, so that RCCL's workgroups cannot hold back the partition levels of the
        # next record's encode (tc_comm_create; TC_COMM_CUS).  TC_BENCH_GATHER=torch: torch.distributed's batch.
        native = os.environ.get("TC_BENCH_GATHER", "native") == "native" and not rehearsal
        if native:
            try:
                gatherer = NativeGather(ctx, pcap, dev, depth=2)
                ok = 1
            except Exception as e:   # noqa: BLE001
                sys.stderr.write("rank %d: native gather unavailable (%s): torch.distributed instead\n" % (rank, e))
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)     # all ranks take the same path
            if int(flag.item()) == 0:
                if gatherer is not None:
                    gatherer.close()
                gatherer, native = None, False
        if gatherer is None:
            gatherer = BlockGather(pcap, xdev, depth=2)
        gatherer.prime()                  # communicator / peer connection set-up, not part of any step
        packed = [torch.empty(pcap, dtype=torch.uint8, device=dev) for _ in range(2)]
    blk = Block()

    def step():
        if gatherer is None:
            blk.nruns = cap
            blk.run_count = d_cnt.data_ptr()
            blk.run_value = d_val.data_ptr()
            rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk))
            if rc != 0:
                raise RuntimeError("tc_encode_dev rc=%d: %s" % (rc, lib.tc_last_error(ctx.handle).decode()))
            return
        if stall:
            time.sleep(10 ** 6)
        buf = packed[gatherer.acquire()]   # the send of the record that last used this buffer has completed
        nb = C.c_uint64(pcap)       # the record as one self-describing container (header + nibble stream)
        rc = lib.tc_encode_container_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.c_void_p(buf.data_ptr()), C.byref(nb))
        if rc != 0:
            raise RuntimeError("tc_encode_container_dev rc=%d: %s" % (rc, lib.tc_last_error(ctx.handle).decode()))
        payload = buf[:nb.value].cpu() if rehearsal else buf
        gatherer.submit([nb.value, 0, 0, 0, 0, n], payload)

    def fence():
        if gatherer is not None:
            gatherer.drain()          # every posted gather completes inside the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, before any warm-up or timed step: the context tries a few placements of its workspace with this
    # record and keeps the fastest (tc_ctx_place_workspace; the partition levels of a long record run at one of
    # two speeds depending on where the workspace lands -- DESIGN.md section 8).  TC_BENCH_PLACE=0: skip.
    placement = None
    if os.environ.get("TC_BENCH_PLACE", "1") != "0" and n >= (1 << 28):
        tries = int(os.environ.get("TC_BENCH_PLACE_TRIES", "5"))
        pms = (C.c_double * 8)()
        pch = C.c_int(-1)
        blk.nruns = cap
        blk.run_count = d_cnt.data_ptr()
        blk.run_value = d_val.data_ptr()
        tp0 = time.perf_counter()
        rc = lib.tc_ctx_place_workspace(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk), tries, pms, C.byref(pch))
        if rc != 0:
            raise RuntimeError("tc_ctx_place_workspace rc=%d: %s" % (rc, lib.tc_last_error(ctx.handle).decode()))
        placement = {"encode_ms_per_placement": [round(x, 2) for x in pms if x > 0], "chosen": pch.value,
                     "seconds": round(time.perf_counter() - tp0, 2)}

    for _ in range(a.warmup):
        step()
    fence()
    if gatherer is not None:
        gatherer.wait_ms.clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    gathered = None
    if rank == 0 and gatherer is not None:
        # outside the timed region: EVERY container gathered for the last record is read back on rank 0 --
        # header, checksum, unpack, full decode -- and compared with the record its rank encoded
        # (regenerated here from the rank's seed): what arrived over RCCL is the right block of bytes.
        last = gatherer.completed[-1]
        assert len(last) == world and all(int(h[0]) == len(p) for h, p in last), "gather shape"
        d_chk = torch.empty(n, dtype=torch.uint8, device=dev)
        d_back = torch.empty(n, dtype=torch.uint8, device=dev)
        vb = Block()
        for r, (h, p) in enumerate(last):
            pd = p.to(dev) if p.device != dev else p
            if pd.data_ptr() % 16:
                pd = pd.clone()
            vb.nruns = cap
            vb.run_count = d_cnt.data_ptr()
            vb.run_value = d_val.data_ptr()
            rc = lib.tc_container_to_block_dev(ctx.handle, C.c_void_p(pd.data_ptr()), int(h[0]), C.byref(vb))
            assert rc == 0, "gathered container of rank %d: rc=%d %s" % (r, rc, lib.tc_last_error(ctx.handle).decode())
            assert int(vb.n) == n, "header of rank %d" % r
            assert lib.tc_decode_dev(ctx.handle, C.byref(vb), C.c_void_p(d_back.data_ptr())) == 0
            assert lib.tc_generate_dev(ctx.handle, 0, 0xC500 + r, n, C.c_void_p(d_chk.data_ptr())) == 0
            torch.cuda.synchronize()
            assert torch.equal(d_back, d_chk), "record of rank %d does not decode to its text" % r
        wm = list(gatherer.wait_ms)
        gathered = {"path": "native" if isinstance(gatherer, NativeGather) else "torch",
                    "fallback_reason": os.environ.get("TC_BENCH_FALLBACK_REASON"),
                    # host time rank 0 spent waiting for posted transfers (tc_comm_wait / the batch's waits), per call of
                    # acquire() / drain() inside the timed region: what of the exchange did NOT hide behind an encode
                    "exchange_wait_ms": {"mean": round(sum(wm) / max(1, len(wm)), 3), "max": round(max(wm) if wm else 0.0, 3),
                                         "calls": len(wm), "last": [round(x, 3) for x in wm[-4:]]},
                    "ranks_in_communicator": dist.get_world_size(),
                    "backend": "tc_comm (RCCL behind the C ABI)" if isinstance(gatherer, NativeGather) else dist.get_backend(),
                    "containers_verified": world, "container_bytes": [int(h[0]) for h, _ in last],
                    "comm_cus": getattr(gatherer, "comm_cus", 0)}
        del d_chk, d_back
    if rank == 0:
        st = ctx.stats()
        total_bytes = n * world * a.steps
        value = total_bytes / dt / 1e6
        A = algorithmic_bytes(st, n)
        launches = int(st.radix_launches)
        roof = None
        if launches:
            avg_ms = st.ms_radix / launches
            per_launch = radix_launch_bytes(st, n)
            ach = per_launch / (avg_ms * 1e-3) / 1e9
            kname = "msd_partition_kernel" if getattr(st, "msd_path", 0) else "radix_pass_kernel"
            roof = {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
                    "launches_per_step": launches, "avg_launch_ms": round(avg_ms, 4),
                    "algorithmic_bytes_per_launch": int(per_launch), "first_pass_builds_keys": bool(st.keygen_fused),
                    "keys_only": bool(getattr(st, "msd_keyonly", 0)),
                    # the SURVEY 8(d) FAMILY FORMULA (prefix doubling with LSD passes over (key, index) pairs) priced at this
                    # step time: comparable across implementations of that family, NOT bytes this implementation moves
                    # (it moves about half: step_traffic_bytes below) and not a bandwidth -- it may exceed the copy rate
                    "pipeline_algorithmic_bytes": A, "pipeline_bytes_per_input_byte": round(A / n, 1),
                    "pipeline_formula_GBps": round(A / (dt / a.steps) / 1e9, 1),
                    "pipeline_formula_note": "SURVEY 8(d) formula bytes / step time: a label for comparison, not measured traffic"}
            # HBM bytes per launch from the PMC counters cannot be collected inside a timed run (rocprofv3
            # --pmc serialises the kernels): the figure is the one recorded by scripts/pmc_traffic.sh for this
            # kernel on this workload (profiles/traffic_latest.json, which names the commit it was taken on)
            pj = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(pj) and n == GIB:
                try:
                    tj = json.load(open(pj))
                    roof["traffic"] = tj.get(kname + "_bytes_per_launch")
                    roof["traffic_source"] = "recorded: profiles/traffic_latest.json (%s), not measured in this run" % tj.get("commit", "?")
                    # the WHOLE step by the counters: every dispatch of one encode call (recorded, as above), against
                    # this run's step time and the 8 TB/s peak
                    if tj.get("step_traffic_bytes"):
                        roof["step_traffic_bytes"] = int(tj["step_traffic_bytes"])
                        roof["step_traffic_GBps"] = round(tj["step_traffic_bytes"] / (dt / a.steps) / 1e9, 1)
                        roof["step_frac"] = round(tj["step_traffic_bytes"] / (dt / a.steps) / 8e12, 4)
                except Exception:
                    pass
        out = {
            "metric": "BWT+MTF+RLE encode MB/s on 1 GiB ACGTN", "value": round(value, 1), "unit": "MB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%d x %d-byte iid ACGTN record(s) (splitmix64 counter generator), fused BWT->MTF->RLE encode, "
                                   "in/out resident in HBM%s" % (world, n, ", + RCCL gather of the encoded-block containers on rank 0 (pipelined)" if world > 1 else ""),
                       "record_bytes": n, "records": world, "parallelism": "record-per-gpu x%d" % world},
            "roofline": roof,
            "gather": gathered,
            "workspace_placement": placement,
            "stages_ms": {"suffix_sort+bwt": round(st.ms_sa, 3), "mtf": round(st.ms_mtf, 3), "rle": round(st.ms_rle, 3),
                          "mtf_rle_one_kernel": bool(st.ms_rle < 0.05 and st.ms_mtf > 0.2),   # (sigma <= 8: "mtf" is then both stages)
                          "rounds": int(st.rounds), "m": [int(st.m[i]) for i in range(st.rounds)],
                          "passes": [int(st.passes[i]) for i in range(st.rounds)], "runs": int(st.runs),
                          "ticket_fallbacks": int(st.ticket_fallbacks)},
        }
        # the extra legs never take the headline with them: a failure is reported in their place
        if world == 1 and os.environ.get("TC_BENCH_CONTAINER", "1") != "0":
            try:
                out["container"] = container_leg(ctx, lib, torch, d_text, n, a.steps, out["ms_per_step"])
            except Exception as e:   # noqa: BLE001
                out["container"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not a.no_host_path and n >= (1 << 24):
            try:
                out["host_path"] = host_path_leg(ctx, lib, torch, d_text, n, out["ms_per_step"])
            except Exception as e:   # noqa: BLE001
                out["host_path"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not a.no_fm and n == GIB:
            del d_cnt, d_val
            try:
                out["fm_count"] = fm_count_leg(ctx, lib, torch, a.no_cpu_baseline)
            except Exception as e:   # noqa: BLE001
                out["fm_count"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not a.no_classes and n == GIB:
            try:
                out["classes"] = classes_leg(ctx, lib, torch)
            except Exception as e:   # noqa: BLE001
                out["classes"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not a.no_cpu_baseline and world == 1:   # the CPU port is timed at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(min(a.cpu_sample, n), seed)
            except Exception as e:   # noqa: BLE001
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
