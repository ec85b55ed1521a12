#!/usr/bin/env python3
"""profiles/<round>/isa/ (AQUA_PROFILE_ROUND, default r05) + profiles/isa_budget.json for the CURRENT build of the benchmarked kernel (no GPU needed).

    make_isa_budget.py <key> <mangled-name substring> stepping=<blocks> stepping_late=<blocks> reseed_pass=<blocks> [...]

  key           row of profiles/isa_budget.json (bench.py: "ns_u8_k8", "ns_f32x2_k8")
  <blocks>      basic-block indices of `isa_budget.py blocks` (lists / ranges) that ONE wavefront executes on the path named:
                  stepping       a stepping wavefront that issues its loads first (wavefront 0 of a block)
                  stepping_late  one that issues them behind its draws (wavefronts 1-3)
                  reseed_pass    one pass of a re-seeding wavefront over its (up to eight) worlds, scan and compaction included
Writes the kernel's listing (<key>.s), its block table (<key>.blocks.txt), the per-path counts (<key>.budget.json) under
profiles/<round>/isa/ and the row bench.py reads (valu issue cycles per stepping wavefront = the average of the two stepping
paths weighted 1 : 3; per re-seeding pass) into profiles/isa_budget.json, tagged with the library's hash.
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "isa_budget.py")
ROUND = os.environ.get("AQUA_PROFILE_ROUND", "r05")          # the listing and counts go to profiles/<round>/isa/
OUT = os.path.join(ROOT, "profiles", ROUND, "isa")


def main(argv):
    if len(argv) < 3:
        sys.exit(__doc__)
    key, name, specs = argv[0], argv[1], argv[2:]
    os.makedirs(OUT, exist_ok=True)
    listing = os.path.join(OUT, key + ".s")
    with open(listing, "w") as f:
        subprocess.check_call([sys.executable, TOOL, "extract", name], stdout=f)
    with open(os.path.join(OUT, key + ".blocks.txt"), "w") as f:
        subprocess.check_call([sys.executable, TOOL, "blocks", listing], stdout=f)
    counts = json.loads(subprocess.check_output([sys.executable, TOOL, "count", listing] + specs))
    sys.path.insert(0, ROOT)
    from aquaticgymenv_amd import _capi
    with open(_capi.LIB_PATH, "rb") as f:
        tag = hashlib.sha256(f.read()).hexdigest()[:16]
    counts["_library_sha16"] = tag
    with open(os.path.join(OUT, key + ".budget.json"), "w") as f:
        json.dump(counts, f, indent=1, sort_keys=True)
    early, late = counts["stepping"]["valu_issue_cycles"], counts.get("stepping_late", counts["stepping"])["valu_issue_cycles"]
    row = {"library_sha16": tag, "source": "profiles/%s/isa/%s.budget.json" % (ROUND, key), "kernel": name,
           "valu_cycles_stepping_wavefront": 0.25 * early + 0.75 * late,
           "valu_cycles_reseed_pass": counts["reseed_pass"]["valu_issue_cycles"],
           "valu_instructions_stepping_wavefront": 0.25 * (counts["stepping"].get("valu", 0) + counts["stepping"].get("lane", 0))
           + 0.75 * (counts.get("stepping_late", counts["stepping"]).get("valu", 0) + counts.get("stepping_late", counts["stepping"]).get("lane", 0))}
    path = os.path.join(ROOT, "profiles", "isa_budget.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        table = {}
    table[key] = row
    with open(path, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print(json.dumps(row, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
