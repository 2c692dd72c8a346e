"""Stamps profile files with the box they were measured on (tools/profile_round.sh): JSON files get a "box" key, text / markdown files a
trailing "# box: ..." line.  usage: stamp_box.py "<box description>" file [file ...]"""
import json
import os
import sys

box = sys.argv[1]
for f in sys.argv[2:]:
    if not os.path.exists(f) or os.path.getsize(f) == 0:
        continue
    if f.endswith(".json"):
        try:
            txt = open(f).read()
            first = txt.splitlines()[0]
            d = json.loads(first if txt.count("\n") <= 1 or not txt.lstrip().startswith("{\n") else txt)
        except Exception:
            try:
                d = json.loads(open(f).read())
            except Exception:
                continue
        d["box"] = box
        json.dump(d, open(f, "w"), indent=None if "metric" in d else 1)
        if "metric" in d:
            open(f, "a").write("\n")
    else:
        with open(f, "a") as h:
            h.write("\n# box: %s\n" % box)
