"""Does the gradient all-reduce of a data-parallel training step start under the backward?  (round-3 verdict item 2c)

Reads the kernel trace of

    RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PREGO_DP_FORCE_COLLECTIVE=1 rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- \
        python3 bench.py --mode train --steps 10 --warmup 3

(one-rank RCCL group, collective path forced: scripts/collect_train_trace.sh) and, for every timed step, lists the kernels that ran
on the COMM stream (a different HIP stream / queue than the backward) with their start relative to the END of the backward's last
kernel on the main stream (layer1's wgrad GEMM).  A sub-bucket whose kernels start before that point travelled under the backward.
Writes a JSON summary (profiles/r04_train_overlap_trace.json).

    python scripts/train_overlap_trace.py OUT_DIR out.json
"""
import csv
import glob
import json
import os
import sys


def main(out_dir, out_json):
    files = glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no *kernel_trace.csv under {out_dir}"
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    skey = "Stream_Id" if "Stream_Id" in rows[0] else qkey
    # a training step ends with the fused AdamW launches; the backward's LAST kernel on the main stream is layer1's wgrad GEMM
    # (the last GEMM before the first adamw kernel of the step)
    adam = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    steps, prev = [], 0
    i = 0
    while i < len(adam):
        j = i
        while j + 1 < len(adam) and adam[j + 1] - adam[j] <= 3:
            j += 1
        steps.append((prev, adam[i], adam[j]))
        prev = adam[j] + 1
        i = j + 1
    main_stream = rows[adam[0]][skey]
    rep = {"kernels": len(rows), "steps_seen": len(steps), "main_stream": main_stream, "per_step": []}
    for lo, a0, a1 in steps[-10:]:                     # the timed steps (warm-up steps come first)
        seg = rows[lo:a0]
        # t = 0: the loss kernel (the backward's launches follow it); "bptt" keeps its name below for the json keys
        bptt = [r for r in seg if "oad_loss_kernel" in r["Kernel_Name"]][-1:]
        if not bptt or not any("gru_bptt" in r["Kernel_Name"] for r in seg):
            continue
        bk = [r for r in seg if "gru_bptt" in r["Kernel_Name"]][-1]
        # the backward's last kernel = the last GEMM on the main stream before AdamW (layer1's wgrad, miniroad.cpp: prego_miniroad_backward)
        mains = [r for r in seg if r[skey] == main_stream and r["s"] >= bptt[0]["s"] and "gemm" in r["Kernel_Name"]]
        last_bwd = max(mains, key=lambda r: r["e"])
        comm = [r for r in seg if r[skey] != main_stream and r["s"] >= bptt[0]["s"]]
        ent = {"backward_last_kernel": last_bwd["Kernel_Name"][:60], "t0": "start of oad_loss_kernel",
               "bptt_kernel_us": [(bk["s"] - bptt[0]["s"]) / 1e3, (bk["e"] - bptt[0]["s"]) / 1e3],
               "backward_end_us": (last_bwd["e"] - bptt[0]["s"]) / 1e3, "comm_stream_kernels": []}
        for r in comm:
            ent["comm_stream_kernels"].append({"name": r["Kernel_Name"][:70], "stream": r[skey],
                                               "start_us": (r["s"] - bptt[0]["s"]) / 1e3,
                                               "start_us_before_backward_end": (last_bwd["e"] - r["s"]) / 1e3,
                                               "dur_us": (r["e"] - r["s"]) / 1e3})
        ent["comm_kernels_started_under_backward"] = sum(1 for k in ent["comm_stream_kernels"] if k["start_us_before_backward_end"] > 0)
        rep["per_step"].append(ent)
    names = {}
    for r in rows:
        names[r["Kernel_Name"][:60]] = names.get(r["Kernel_Name"][:60], 0) + 1
    rep["rccl_kernels"] = {k: v for k, v in names.items() if "nccl" in k.lower() or "rccl" in k.lower()}
    rep["note"] = ("one-rank communicator: RCCL's in-place all-reduce of a single rank launches no kernel (rccl_kernels is empty); what the "
                   "comm stream shows per sub-bucket is the averaging kernel (x 1 / world) that is stream-ordered BEHIND the ncclAllReduce call, so "
                   "its start is the earliest the collective could have completed.  Buckets 0 (head) and 1 (GRU) start before the backward's last "
                   "GEMM ends; bucket 2 (layer1 + LayerNorm) is final only when the backward is.")
    st = [[k["start_us_before_backward_end"] for k in e["comm_stream_kernels"][:3]] for e in rep["per_step"] if len(e["comm_stream_kernels"]) >= 3]
    if st:
        rep["median_start_us_before_backward_end"] = [sorted(x[i] for x in st)[len(st) // 2] for i in range(3)]
    json.dump(rep, open(out_json, "w"), indent=1)
    print(json.dumps(rep, indent=1)[:6000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
