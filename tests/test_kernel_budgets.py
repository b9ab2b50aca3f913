"""CPU-only: register / scratch budgets of the kernels the benchmark path stands on, read from hipcc's
resource-usage remarks (a cross-compile, no GPU).  A branch added to msd_finish_kernel once raised it from
80 to 123 VGPRs (6 -> 4 waves per SIMD, +3 ms on the 1 GiB step) and went unnoticed for a few commits: this
pins the budgets the measured numbers were taken with."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "text-compression_amd")

# mangled-name fragment -> (max VGPRs, max scratch bytes per lane, min waves per SIMD)
BUDGETS = {
    "msd_finish_kernelILi256ELi8ELi4ELi1ELb0ELb1EE": (96, 0, 5),
    "msd_finish_kernelILi512ELi12ELi1ELi5ELb1ELb1EE": (128, 0, 4),
    "msd_partition_kernelILb0ELb1EE": (128, 0, 4),
    "msd_partition_kernelILb1ELb1EE": (128, 16, 4),
    "msd_count_kernelILb0ELb1EE": (64, 0, 4),
    "13finish_kernel10FinishArgs": (96, 0, 5),
    "rle_encode_idx_kernelIhE": (128, 0, 4),
    "mtf_nib_apply_kernelI6BwtAccLb1EhLb0EE": (80, 0, 4),     # (round 4: 64 -> 76 with the far backward scan's 16-byte loads; still 4 waves)
    "mtf_nib_apply_kernelI6BwtAccLb1EhLb1EE": (80, 0, 4),     # round 3: the list in 32 bits; a tile's 8 loads in flight together
    "mtf_ts_apply_kernelI6BwtAccLi5EE": (64, 0, 8),
    "radix_pass_kernelILb0ELb0ELi1ELb0ELb0EE": (160, 0, 3),
    "14rle_nib_kernel10RleNibArgs": (128, 0, 4),          # round 3: the fused RLE -> wire-format kernel (2 x 512 threads per CU)
    "15pack_nib_kernel11PackNibArgs": (96, 0, 5),
    "fm_count_kernelILb1EE": (64, 0, 8),                   # round 3: two symbols per lookup
    "rle_blk_kernelILb1EE": (96, 0, 4),                    # round 3: blocked RLE of the byte-wide index stream
    "msd_partition_kernelILb0ELb0EE": (128, 0, 4),         # round 3: key-only levels
    "msd_partition_kernelILb1ELb0EE": (128, 16, 4),
    "msd_finish_kernelILi256ELi8ELi4ELi1ELb0ELb0EE": (80, 0, 6),
    "tied_probe_kernelILi3EE": (128, 0, 3),               # (LDS holds 3 workgroups per CU; the next tile's text sits in 20 registers)
    "msd_finish_ko_kernelILi3EE": (80, 0, 6),              # round 3: equal-mass bins, prefetched keys.  NO scratch: a build
                                                           # of it that spilled (12 bytes) gave wrong tied sets on the GPU
    "mtf_rle_kernelILb0EE": (96, 0, 4),                    # round 3: MTF and RLE of a small-alphabet record in one kernel (run arrays)
    "mtf_rle_kernelILb1EE": (128, 0, 4),                   # ... writing the container's nibble stream (LDS holds 4 workgroups per CU)
}


def test_kernel_register_budgets():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                          "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"),
                          "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/libtextcomp_budget.so",
                          os.path.join(PKG, "csrc", "textcomp.hip")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    seen = {}
    for blk in out.stderr.split("Function Name: ")[1:]:
        name = blk.split()[0]
        v = int(re.search(r"VGPRs: (\d+)", blk).group(1))
        s = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
        o = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", blk).group(1))
        for frag in BUDGETS:
            if frag in name:
                seen[frag] = (v, s, o)
    for frag, (mv, ms, mo) in BUDGETS.items():
        assert frag in seen, "kernel not found: " + frag
        v, s, o = seen[frag]
        assert v <= mv and s <= ms and o >= mo, (frag, "VGPRs %d (<= %d), scratch %d (<= %d), waves/SIMD %d (>= %d)" % (v, mv, s, ms, o, mo))
