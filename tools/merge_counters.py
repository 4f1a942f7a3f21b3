"""Merge gpurun_out/<tag>/counters.json (tools/prof.sh) into profiles/traffic.json and profiles/valu.json under a config
name:   python tools/merge_counters.py gpurun_out/r02x C2"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, cfg = sys.argv[1], sys.argv[2]
c = json.load(open(os.path.join(src, "counters.json")))
for name, key in (("traffic.json", "traffic"), ("valu.json", "valu")):
    path = os.path.join(ROOT, "profiles", name)
    cur = json.load(open(path)) if os.path.exists(path) else {}
    if key == "valu":
        cur.setdefault("_note", "vector-ALU counters per launch from rocprofv3 --pmc passes (tools/prof.sh): valu_busy = SQ_ACTIVE_INST_VALU "
                                "(quad-cycles) * 4 / (1024 SIMDs * kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs")
    cur[cfg] = c[key]
    cur.setdefault("_source", {})[cfg] = os.path.basename(os.path.normpath(src))
    # sha256 of the rasterizer sources (topo4d_amd.build.raster_source_sha256) when the counters were taken: bench.py compares it with the source it runs
    cur.setdefault("_kernel_source_sha256", {})[cfg] = c.get("_kernel_source_sha256")
    json.dump(cur, open(path, "w"), indent=1)
    print("updated", path)
