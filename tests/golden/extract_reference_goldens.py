#!/usr/bin/env python3
"""Extracts the known-answer vectors the reference's own tests hold for this path into
tests/golden/reference_goldens.json.  Run in the build container (needs /root/reference):

    python tests/golden/extract_reference_goldens.py

Sources (tracel-ai/cubecl checkout):
  crates/cubecl-core/src/runtime_tests/cmma.rs   test_simple_1_expected (:552-576),
                                                 test_simple_tf32 expected (:868-889),
                                                 test_cmma_strided expected (:932-1005)
Only literal numbers are copied (test data, not code); inputs are re-stated as formulas in
tests/test_oracle_golden.py with the reference line they come from.
"""
import json
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "reference_goldens.json"


def numbers(block: str):
    return [float(tok) for tok in re.findall(r"-?\d+(?:\.\d*)?", block)]


def literal_after(text: str, marker: str, opener: str):
    start = text.index(marker)
    i = text.index(opener, start) + len(opener)
    j = text.index("]", i)
    return numbers(text[i:j])


def main():
    cmma = (REF / "crates/cubecl-core/src/runtime_tests/cmma.rs").read_text()
    goldens = {
        "cmma_simple_1_f16_16x16x16_nt": literal_after(cmma, "pub fn test_simple_1_expected()", "vec!["),
        "cmma_simple_tf32_16x16x8_nn": literal_after(cmma, "pub fn test_simple_tf32<", "let expected = ["),
        "cmma_strided_f16_16x16x16_nt": literal_after(cmma, "pub fn test_cmma_strided<", "let expected = ["),
    }
    for k, v in goldens.items():
        assert len(v) == 256, (k, len(v))
    OUT.write_text(json.dumps(goldens, indent=0, separators=(",", ":")))
    print("wrote", OUT, {k: len(v) for k, v in goldens.items()})


if __name__ == "__main__":
    main()
