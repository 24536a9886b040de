"""Generates tests/golden/hc_dict_digests.json from the compiled reference (oracle/_ref = liblz4 1.10.0 built from
/root/reference): SHA-256 of every record plz4 would write at levels 2..12 for (a) a linked frame with a dictionary, (b) a
linked frame without one, (c) independent blocks with a dictionary.  Run from the repo root in the build container."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import hcdict                      # noqa: E402
from orclib import Oracle, Ref     # noqa: E402


def main():
    ref, orc = Ref(), Oracle()
    user, bsz, frame, indie = hcdict.golden_inputs()
    out = {"reference": "liblz4 1.10.0 LZ4_compress_HC_continue as driven by clz4.go (StreamCtxHC / StreamLinkedCtxHC)",
           "src_sha": hashlib.sha256(b"".join(b.tobytes() for b in [user] + frame + indie)).hexdigest(), "levels": {}}
    for lvl in range(2, 13):
        e = {}
        for name, blocks, linked, dct in (("linked_dict", frame, True, user), ("linked", frame, True, None), ("indie_dict", indie, False, user)):
            recs, rets = hcdict.ref_records(ref, orc, blocks, bsz, lvl, linked, dct)
            e[name] = {"ret": rets, "sha": [hashlib.sha256(r).hexdigest() for r in recs]}
        out["levels"][str(lvl)] = e
    with open(os.path.join(ROOT, "tests", "golden", "hc_dict_digests.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("written", len(out["levels"]), "levels")


if __name__ == "__main__":
    main()
