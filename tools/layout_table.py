#!/usr/bin/env python3
"""What a picture layout costs: slice_data bytes (the device's arithmetic coder) and PSNR of the same 1080p pictures coded as one tile per picture (the reference cfg's
layout), one tile + WaveFrontSynchro, 4 x 2 tiles and 15 x 9 tiles (one CTU per tile: bench.py's default), at QP 22 / 27 / 32 / 37 with the full tool set, and the
Bjontegaard delta rate of every layout against the cfg's.  All sixteen encodes are submitted at once on streams of their own (vvcx_submit_ctus): the one-tile pictures
are single 135-CTU chains that take minutes whatever else runs.

  python tools/layout_table.py [--frames 8] [--out profiles/r04_layout_table.json]      (needs a GPU; ~15 minutes)
"""
import argparse
import importlib
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bd_rate import bd_rate      # noqa: E402

LAYOUTS = (("1x1", 1, 1, False), ("1x1+wpp", 1, 1, True), ("4x2", 4, 2, False), ("15x9", 15, 9, False))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--qps", type=str, default="22,27,32,37")
    ap.add_argument("--layouts", type=str, default=",".join(l[0] for l in LAYOUTS))
    ap.add_argument("--out", type=str, default=None)
    a = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # the encodes of a run overlap on streams of their own: do not let them share the default four hardware queues
    import torch
    pkg = importlib.import_module("reduce-complexity-for-intra-coding-of-vvc_amd")
    W, H = a.width, a.height
    ctus = ((W + 127) // 128) * ((H + 127) // 128)
    frames = [pkg.synth_frame(W, H, poc, 8, 1000 + poc, chroma_texture=0.5) for poc in range(a.frames)]
    org = [[torch.from_numpy(p).cuda() for p in f] for f in frames]
    jobs = []
    for qp in [int(v) for v in a.qps.split(",")]:
        for name, tc, tr, wpp in LAYOUTS:
            if name not in a.layouts.split(","):
                continue
            tools = 0xfff | (pkg.TOOL_WPP if wpp else 0)
            sp = pkg.slice_params(qp, dep_quant=True)
            enc = pkg.VvcxEncoder(W, H, 8, tile_cols=tc, tile_rows=tr, tools=tools, max_frames=a.frames, emit_payload=True)
            enc.set_slice(sp["qp"], sp["qp_c"], sp["lam"], sp["dist_weight"])
            rec = [[torch.zeros_like(t) for t in f] for f in org]
            enc.bind_frames([([t.data_ptr() for t in o], [t.data_ptr() for t in r], [t.shape[1] for t in o]) for o, r in zip(org, rec)])
            stream = torch.cuda.Stream()
            tasks = [(f, c) for f in range(a.frames) for t in range(tc * tr) for c in pkg.tile_ctus((W + 127) // 128, (H + 127) // 128, tc, tr, t)]
            if wpp:                                        # rows of a picture top-down, CTU by CTU along the anti-diagonals (no row ahead of the row above it)
                cw, chh = (W + 127) // 128, (H + 127) // 128
                tasks = [(f, y * cw + x) for f in range(a.frames) for y in range(chh) for x in range(cw)]
            n = enc.submit_ctus(tasks, stream.cuda_stream)
            jobs.append(dict(qp=qp, layout=name, tc=tc, tr=tr, enc=enc, rec=rec, n=n, stream=stream, t0=time.time(), done=False))
    t0 = time.time()
    while not all(j["done"] for j in jobs):
        time.sleep(20)
        for j in jobs:
            if not j["done"] and j["enc"].poll_ctus():
                j["enc"].wait_ctus(j["n"]); j["done"] = True; j["seconds"] = time.time() - j["t0"]
        print("[layout] %4.0f s: %d of %d encodes finished" % (time.time() - t0, sum(j["done"] for j in jobs), len(jobs)), flush=True)
    rows = []
    for j in jobs:
        enc = j["enc"]
        nbytes = sum(len(enc.get_payload(f, t)) for f in range(a.frames) for t in range(j["tc"] * j["tr"]))
        sse = [0.0, 0.0, 0.0]; npx = [0, 0, 0]
        for o, r in zip(org, j["rec"]):
            for c in range(3):
                d = o[c].to(torch.float64) - r[c].to(torch.float64)
                sse[c] += float((d * d).sum().item()); npx[c] += d.numel()
        psnr = [10 * np.log10(255.0 ** 2 / (sse[c] / npx[c])) for c in range(3)]
        rows.append(dict(qp=j["qp"], layout=j["layout"], bits=8 * nbytes, psnr_y=psnr[0], psnr_u=psnr[1], psnr_v=psnr[2], psnr_yuv=(6 * psnr[0] + psnr[1] + psnr[2]) / 8,
                         seconds_alone_or_shared=j["seconds"], kernel_ms=enc.last_kernel_ms(), ctus=a.frames * ctus))
        print("QP %d %-8s %10d bits  %.3f dB Y  %.3f dB YUV  kernel %.1f s" % (j["qp"], j["layout"], 8 * nbytes, psnr[0], rows[-1]["psnr_yuv"], enc.last_kernel_ms() / 1e3), flush=True)
        enc.close()
    out = {"workload": "%dx%d 8-bit 4:2:0, %d synthetic frames, tools 0xfff, QP %s" % (W, H, a.frames, a.qps), "rows": rows, "bd_rate_vs_1x1_percent": {}}
    ref = sorted([r for r in rows if r["layout"] == "1x1"], key=lambda r: r["qp"])
    if len(ref) >= 4:
        for name in sorted({r["layout"] for r in rows} - {"1x1"}):
            t = sorted([r for r in rows if r["layout"] == name], key=lambda r: r["qp"])
            out["bd_rate_vs_1x1_percent"][name] = {"y": bd_rate([r["bits"] for r in ref], [r["psnr_y"] for r in ref], [r["bits"] for r in t], [r["psnr_y"] for r in t]),
                                                   "yuv": bd_rate([r["bits"] for r in ref], [r["psnr_yuv"] for r in ref], [r["bits"] for r in t], [r["psnr_yuv"] for r in t]),
                                                   "bits_ratio_per_qp": [t_["bits"] / r_["bits"] for r_, t_ in zip(ref, t)]}
    print(json.dumps(out["bd_rate_vs_1x1_percent"]))
    if a.out:
        path = os.path.join(ROOT, a.out) if not os.path.isabs(a.out) else a.out
        if os.path.exists(path):                            # a second run with other --layouts adds its rows (the one-tile chains take a call of their own)
            old = json.load(open(path))
            rows = [r for r in old["rows"] if (r["qp"], r["layout"]) not in {(q["qp"], q["layout"]) for q in rows}] + rows
            out["rows"] = rows
            ref = sorted([r for r in rows if r["layout"] == "1x1"], key=lambda r: r["qp"])
            for name in sorted({r["layout"] for r in rows} - {"1x1"}):
                t = sorted([r for r in rows if r["layout"] == name], key=lambda r: r["qp"])
                if len(ref) >= 4 and len(t) == len(ref):
                    out["bd_rate_vs_1x1_percent"][name] = {"y": bd_rate([r["bits"] for r in ref], [r["psnr_y"] for r in ref], [r["bits"] for r in t], [r["psnr_y"] for r in t]),
                                                           "yuv": bd_rate([r["bits"] for r in ref], [r["psnr_yuv"] for r in ref], [r["bits"] for r in t], [r["psnr_yuv"] for r in t]),
                                                           "bits_ratio_per_qp": [t_["bits"] / r_["bits"] for r_, t_ in zip(ref, t)]}
        json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
