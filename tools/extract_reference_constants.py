#!/usr/bin/env python3
"""Extract the literal known-answer constants the reference pins for this path
into tests/golden/reference_constants.json  (data only -- numbers, no source).

Sources (all under /root/reference/mopro-msm/src/msm/metal_msm/, present only
in the dev container; this script never runs on the GPU box):
  shader/constants.metal:9-282   N0, NSAFE, SLACK, BARRETT_MU, modulus p, R,
                                 identity (1,1,0), generator (1,2,1) and their
                                 Montgomery images, as 16 x 16-bit limbs
  utils/mont_params.rs:116-122   R^-1 mod p (decimal), n0 = 25481
  utils/barrett_params.rs:25-28  Barrett mu (decimal)
"""
import json
import os
import re

REF = "/root/reference/mopro-msm/src/msm/metal_msm"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                   "reference_constants.json")


def limbs16_to_int(limbs):
    v = 0
    for i, l in enumerate(limbs):
        v |= l << (16 * i)
    return v


def main():
    txt = open(os.path.join(REF, "shader/constants.metal")).read()
    out = {"_source": "constants.metal / mont_params.rs / barrett_params.rs literals of the reference"}
    for m in re.finditer(r"#define\s+(\w+)\s+(\d+)", txt):
        out[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"constant uint32_t (\w+)\[\w+\] = \{([^}]*)\}", txt):
        limbs = [int(x) for x in m.group(2).replace("\n", " ").split(",") if x.strip()]
        out[m.group(1)] = {"limbs16": limbs, "hex": hex(limbs16_to_int(limbs))}
    mp = open(os.path.join(REF, "utils/mont_params.rs")).read()
    out["RINV_DECIMAL"] = re.search(r'rinv == BigUint::from_str\(\s*"(\d{60,})"', mp).group(1)
    out["N0_TEST"] = int(re.search(r"n0 == (\d+)u32", mp).group(1))
    bp = open(os.path.join(REF, "utils/barrett_params.rs")).read()
    out["BARRETT_MU_DECIMAL"] = re.search(r'"(\d{60,})"', bp).group(1)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, "with", len(out), "entries")


if __name__ == "__main__":
    main()
