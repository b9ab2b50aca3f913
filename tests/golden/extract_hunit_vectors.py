#!/usr/bin/env python3
"""Transcribe the reference's own known-answer DATA into tests/golden/hunit_vectors.json.

Reads (as text -- nothing is executed) the literal test vectors the reference
holds for this path and writes them as JSON:

  src/Data/RLE.hs:279-311   rle1, s1, rle2, s2         (HUnit cases RLE.hs:313-320)
  src/Data/MTF.hs:287-299   "aaabbbccc" <-> MTF vector  (HUnit cases)
  src/Data/FMIndex/Internal.hs:49-113  abracadabra doc tables (L, C[c], Occ(c,k))

Only inputs and expected outputs are copied; no reference code.  Runs in the
build container only (/root/reference does not exist on the GPU box); the JSON it
produced is committed next to it.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def maybe_list(src):
    """Parse a Haskell list of `Just "x"` / `Nothing` into [str|None]."""
    out = []
    for m in re.finditer(r'Just\s+"((?:[^"\\]|\\.)*)"|Nothing', src):
        out.append(None if m.group(0) == "Nothing" else m.group(1))
    return out


def main():
    rle = open(os.path.join(REF, "src/Data/RLE.hs")).read()
    mtf = open(os.path.join(REF, "src/Data/MTF.hs")).read()
    fmi = open(os.path.join(REF, "src/Data/FMIndex/Internal.hs")).read()

    rle1 = maybe_list(re.search(r"rle1 = RLE \(fromList \[(.*?)\]\)", rle, re.S).group(1))
    rle2 = maybe_list(re.search(r"rle2 = RLE \(fromList \[(.*?)\]\)", rle, re.S).group(1))
    s1 = re.search(r'^s1 = "(.*)"$', rle, re.M).group(1)
    s2 = re.search(r'^s2 = "(.*)"$', rle, re.M).group(1)

    m = re.search(r'MTF \(\[([0-9,]+)\],\s*\[(.*?)\]\)\)\s*\(textToBWTToMTFB "(\w+)"\)', mtf, re.S)
    mtf_idx = [int(v) for v in m.group(1).split(",")]
    mtf_list = maybe_list(m.group(2))
    mtf_in = m.group(3)

    # abracadabra doc tables
    text = re.search(r'Given the following input, "(\w+)"', fmi).group(1)
    L = re.search(r'C\[c\] of "([^"]+)"', fmi).group(1)
    csyms = [c.strip() for c in re.search(r"^-- \| c\s+\|(.*)\|\s*$", fmi, re.M).group(1).split("|")]
    cvals = [int(v) for v in re.search(r"^-- \| C\[c\] \|(.*)\|\s*$", fmi, re.M).group(1).split("|")]
    occ = {}
    occ_block = fmi[fmi.index("Occ(c,k) of"):fmi.index("Keep in mind")]
    for line in occ_block.splitlines():
        mm = re.match(r"-- \| (\S) \|((?:\s*\d+\s*\|)+)\s*$", line)
        if mm and not mm.group(1).isdigit():
            occ[mm.group(1)] = [int(v) for v in mm.group(2).split("|") if v.strip()]
    assert len(rle2) == 174 and len(s2) == 565 and len(occ) == 6, (len(rle2), len(s2), len(occ))

    out = {
        "source": "Matthew-Mosior/text-compression v0.1.0.25: RLE.hs:279-311, MTF.hs:287-299, "
                  "FMIndex/Internal.hs:49-113 (data only)",
        "rle": [{"text": s1, "rle": rle1}, {"text": s2, "rle": rle2}],
        "mtf": [{"text": mtf_in, "indices": mtf_idx, "final_list": mtf_list}],
        "fmindex_doc": {"text": text, "L": L, "C": dict(zip(csyms, cvals)), "Occ": occ},
    }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hunit_vectors.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
