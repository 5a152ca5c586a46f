"""Rebuild profiles/<round>_pmc_valu.json and <round>_pmc_traffic.json (round = argv[1], default r02) from the four rocprofv3
PMC passes of tools/run_pmc.sh over tools/extract_only.py (rocpd databases): gpurun_out/pmc_x_{insts,active} (64-frame batch) and
pmc_x_{fetch,write} (256-frame batch)."""
import sqlite3, collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r02"
def load(tag):
    c = sqlite3.connect(os.path.join(ROOT, "gpurun_out", "pmc_x_" + tag, "run_results.db"))
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    for did, name, cn, val in c.execute("select dispatch_id,kernel_name,counter_name,value from counters_collection"):
        k = name.split("(")[0].replace("orbhip::", "").replace("void ", "").split("<")[0]
        a = acc[k][cn]; a[0] += val; a[1].add(did)
    return {k: {cn: v[0] / len(v[1]) for cn, v in d.items()} for k, d in acc.items()}      # mean per launch (rows of one dispatch summed)
ins, act, fe, wr = load("insts"), load("active"), load("fetch"), load("write")
names = ["k_resize", "k_fast_cells", "k_octree", "k_blur7", "k_describe"]
old = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_valu.json")))        # (the note describing the method)
valu = {"note": old["note"], "frames_per_launch": 64, "kernels": {}}
for k in names:
    i, a = ins[k], act[k]
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    launches = 7 if k == "k_resize" else 1
    valu["kernels"][k] = {
        "waves": i["SQ_WAVES"], "valu_insts": i["SQ_INSTS_VALU"], "salu_insts": i["SQ_INSTS_SALU"], "lds_insts": i["SQ_INSTS_LDS"],
        "vmem_rd_insts": i["SQ_INSTS_VMEM_RD"], "vmem_wr_insts": i["SQ_INSTS_VMEM_WR"], "kernel_cycles": cyc,
        "valu_insts_per_wave": i["SQ_INSTS_VALU"] / i["SQ_WAVES"],
        "valu_busy_frac": i["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc,
        "avg_waves_per_simd": a["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / cyc if "SQ_WAVE_CYCLES" in a else None,
        "wait_any_frac_of_wave_cycles": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in a else None,
        "lds_bank_conflict_cycles": a.get("SQ_LDS_BANK_CONFLICT", 0.0),
    }
json.dump(valu, open(os.path.join(ROOT, "profiles", RND + "_pmc_valu.json"), "w"), indent=1)
oldt = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
tr = {"note": oldt["note"], "frames_per_launch": 256, "kernels": {}}
for k in names:
    mult = 7 if k == "k_resize" else 1                      # 7 pyramid launches per batch
    f, w = fe[k]["FETCH_SIZE"] * mult, wr[k]["WRITE_SIZE"] * mult
    tr["kernels"][k] = {"FETCH_SIZE_KB_per_batch": f, "WRITE_SIZE_KB_per_batch": w, "hbm_bytes_per_frame": (f + w) * 1024.0 / 256.0}
json.dump(tr, open(os.path.join(ROOT, "profiles", RND + "_pmc_traffic.json"), "w"), indent=1)
for k in names:
    v = valu["kernels"][k]
    print("%-14s VALU/wave %.0f  busy %.3f  waves/SIMD %.2f  HBM B/frame %.0f" % (k, v["valu_insts_per_wave"], v["valu_busy_frac"], v["avg_waves_per_simd"] or 0, tr["kernels"][k]["hbm_bytes_per_frame"]))
