"""Development tool: per-kernel means of rocprofv3 --pmc counter passes (any counters).
usage: summarise_counters.py out.json pass1.csv [pass2.csv ...]   -- one CSV per rocprofv3 pass."""
import csv
import json
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").replace("stereo::", "").split("(")[0][:80]


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = {}
    for path in files:
        with open(path) as f:
            for row in csv.DictReader(f):
                k, c = short(row["Kernel_Name"]), row["Counter_Name"]
                d = acc.setdefault(k, {}).setdefault(c, {})
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    res = {}
    for k, cs in acc.items():
        if "trws_" not in k and "qpbo_maxflow" not in k and "ncc_" not in k:
            continue
        res[k] = {c: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v)} for c, v in cs.items()}
        m = {c: x["mean_per_launch"] for c, x in res[k].items()}
        derived = {}
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
            for c in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if c in m:
                    derived[c + "/SQ_WAVE_CYCLES"] = m[c] / m["SQ_WAVE_CYCLES"]
        if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"]:
            derived["valu_instructions_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
        if "SQ_LDS_BANK_CONFLICT" in m and "SQ_ACTIVE_INST_LDS" in m and m["SQ_ACTIVE_INST_LDS"]:
            derived["lds_bank_conflict_cycles/lds_active_cycles"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_ACTIVE_INST_LDS"]
        if derived:
            res[k]["derived"] = derived
    json.dump({"note": "means per launch over all dispatches of the kernel; SQ counters are summed over the shader engines "
                       "as rocprofv3 reports them; one rocprofv3 run per counter group, never combined with trace domains "
                       "other than --kernel-trace", "per_kernel": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
