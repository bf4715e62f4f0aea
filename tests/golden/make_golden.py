#!/usr/bin/env python3
"""Regenerate tests/golden/*.f32 (packed fp32 framebuffers, [x][z][3]).

The images are produced by the CPU oracle (oracle/rt_oracle.c) and are kept
only if their SHA-256 equals the digest SURVEY.md Appendix D recorded for the
same scene/size/depth from the survey's build of the reference.  The reference
itself cannot be built in this round (missing headers), so it is not run here.
Also writes extra.json: digests of larger oracle renders used by GPU tests.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402

for key in ("b64d4", "g32_64d4", "g16_64d8"):
    name, W, H, depth, digest = oracle_lib.SURVEY_PINS[key]
    img = oracle_lib.OracleScene.named(name).render(W, H, depth)
    assert oracle_lib.sha256(img) == digest, key
    img.tofile(os.path.join(HERE, key + ".f32"))
    print("wrote", key, digest)

extra = {"_about": "SHA-256 of oracle renders (oracle/rt_oracle.c). ORACLE-ONLY pins: SURVEY.md Appendix D's digests cover the built-in "
                   "scene and the NO-shadow grids; the shadowed grid32 / grid16 scenes (the ones bench.py and the 4096^2 tests render) and "
                   "twomirrors have no digest from the reference, so these values pin the GPU path to the oracle, not the oracle to the reference."}
for name, W, H, depth in (("grid32", 64, 64, 4), ("grid16", 64, 64, 8), ("twomirrors", 48, 48, 6),
                          ("builtin", 500, 504, 50), ("grid32", 256, 256, 4), ("grid16", 256, 256, 8)):
    img = oracle_lib.OracleScene.named(name).render(W, H, depth)
    extra[f"{name}_{W}x{H}_d{depth}"] = oracle_lib.sha256(img)
json.dump(extra, open(os.path.join(HERE, "extra.json"), "w"), indent=1, sort_keys=True)
print(extra)
