"""Oracle digests of records AWAY from iid ACGTN, at sizes where the library's other paths change regime
(CPU only; test infrastructure).  The oracle (oracle/tc_oracle.c) encodes each record of tests/classgen.py
and its digest -- primary, sigma, final MTF list, number of runs, max run, position-dependent 64-bit
checksums of the last column, run_count[] and run_value[] -- goes to tests/golden/classes_digest.json;
tests/test_gpu_classes_digest.py regenerates the same bytes on the GPU box's host (numpy, integer
arithmetic only), encodes them on the device and compares digest for digest.

What each record is there for (src/Data/BWT.hs:68-70 accepts any ByteString; BWT/Internal.hs:110-134,
MTF/Internal.hs:128-175):
  zipf_words  2^27  nearly every suffix tied after round 0: full LSD path, dense ranks stored by regions
                    (rank_bin_kernel, from 2^25 members on), lane-chunk MTF at sigma = 28
  bytes256    2^26  sigma = 257: timestamp MTF with the sentinel split; 2-byte run format
  ascii96     2^26  sigma = 96: timestamp MTF (default beyond 64 symbols)
  acgt4       2^28  MSD round 0 with the BIG finish instance chosen by the estimate (4-letter DNA: ~1024
                    suffixes per level-3 bucket), sorted-key rank lookups of the doubling rounds
  genome_like 2^27  repeat-rich DNA: the collision sample sends it from the MSD way to the LSD way; over-long
                    buckets -> finish_fix tied groups; prefix doubling on a large tied set

Run once:  python tests/long/classes_digest.py [name:log2n ...]     (minutes per record, <= 8 GB)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import classgen  # noqa: E402
from parity_digest import digest_text  # noqa: E402

DEFAULT = [("bytes256", 26), ("ascii96", 26), ("zipf_words", 27), ("genome_like", 27), ("acgt4", 28)]

if __name__ == "__main__":
    todo = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]] or DEFAULT
    out = os.path.join(ROOT, "tests", "golden", "classes_digest.json")
    for name, lg in todo:
        n = 1 << lg
        print("%s n=2^%d" % (name, lg), flush=True)
        d = digest_text(classgen.make(name, n), log=lambda s: print(s, flush=True))
        d["class"] = name
        res = json.load(open(out)) if os.path.exists(out) else {}   # (several of these may run side by side)
        res["_doc"] = ("digests of the ORACLE's BWT->MTF->RLE encode of tests/classgen.py records; written by "
                       "tests/long/classes_digest.py; checksum64 = checksum64_kernel of csrc/textcomp.hip")
        res["%s_n%d" % (name, n)] = d
        json.dump(res, open(out, "w"), indent=1, sort_keys=True)
