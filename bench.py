#!/usr/bin/env python3
"""bench.py — CTUs/sec of the intra CU-partition RDO hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of 1920x1080 8-bit 4:2:0 All-Intra frames (135 CTUs of 128x128
each) at QP 32, full RDO (no early termination) with the tool subset built so far (see config.tools).  Every frame is
cut into a uniform 15x9 tile grid so that every CTU is an independent stream (one workgroup per stream, SURVEY.md
§8e); the batch holds enough frames for ~4 full waves of resident streams (30 frames on a 256-CU part; fewer, down to two
waves, when steps + warmup would not fit the run budget at that size; --frames to override); original planes are resident in HBM before the timed region.  With --gpus N every rank encodes its own
frames per step (weak scaling, no data-path collective; frames are independent in All-Intra).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against HBM bandwidth with the
algorithmic bytes of SURVEY.md §8d (49 152 B per 8-bit CTU) and adds the VALU issue fraction from the newest committed
PMC summary (the kernel is instruction-issue bound, not HBM bound); `cpu_baseline` times the CPU oracle (a port of the
reference path, oracle/) on a bounded sample of the same workload on this host and quotes the reference encoder's own
measured rate (one core of the authoring container) beside it.

`--gpus N` without a torchrun environment launches N ranks itself (before anything touches the GPU) and exits with
their code.  With N > 1 every rank also emits its tiles' slice_data payloads and the final bitstream gather to rank 0
(two RCCL collectives, sharding.gather_payloads) is timed after the steps (`gather` object).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

B_CTU_8BIT = 49152          # 2 x 1.5 x 128 x 128 x 1 byte  
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


RUN_BUDGET_S = 380.0          # what all timed + warm-up steps of one run may take (auto batch size only)
BUDGET_CTUS_PER_S = 150.0     # rate assumed for that (below the measured rate of the full tool set: 160-173 CTU/s in round 4)


def pmc_traffic(workload):
    """HBM bytes per launch from the newest committed PMC summary of this exact workload (hardware counters cannot be
    read from inside the process; tools/gpu_profile_round.sh collects FETCH_SIZE / WRITE_SIZE in separate rocprofv3
    --pmc passes and tools/rocpd_summary.py applies the gfx950 correction).  None if no summary matches."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        if d.get("workload") == workload and "hbm_traffic_bytes_per_launch" in d:
            return d["hbm_traffic_bytes_per_launch"], os.path.relpath(p, ROOT), {"read": d.get("read_bytes_per_launch"), "write": d.get("write_bytes_per_launch")}
    return None, None, None


def pmc_valu(workload):
    """VALU issue fraction of the dominant kernel from the same PMC summary: SQ_INSTS_VALU per launch x 4 cycles (one wave64 VALU instruction
    occupies its SIMD for 4 cycles) / (kernel cycles x 1024 SIMDs).  None if no summary matches."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        pl = d.get("per_launch_avg", {})
        if d.get("workload") == workload and "SQ_INSTS_VALU" in pl and "SQ_BUSY_CYCLES" in pl:
            # SQ_BUSY_CYCLES is summed over the 32 shader engines' SQs x XCDs as collected; the launch duration of the same run is the robust denominator
            ms = d.get("kernel_ms")
            out = {"insts_valu_per_launch": pl["SQ_INSTS_VALU"], "wave_cycles_per_launch": pl.get("SQ_WAVE_CYCLES"), "wait_any_per_launch": pl.get("SQ_WAIT_ANY"),
                   "source": os.path.relpath(p, ROOT)}
            if ms:
                out["issue_frac"] = pl["SQ_INSTS_VALU"] * 4.0 / (ms * 1e-3 * 2.4e9 * 1024)
            if pl.get("SQ_WAVE_CYCLES") and pl.get("SQ_WAIT_ANY") is not None:
                out["wait_any_frac"] = pl["SQ_WAIT_ANY"] / pl["SQ_WAVE_CYCLES"]
            if pl.get("SQ_LDS_IDX_ACTIVE") and pl.get("SQ_LDS_BANK_CONFLICT") is not None:      # cycles lost to bank conflicts per cycle the LDS index unit is busy
                out["lds_bank_conflict_frac"] = pl["SQ_LDS_BANK_CONFLICT"] / pl["SQ_LDS_IDX_ACTIVE"]
            if pl.get("SQ_ACTIVE_INST_VALU") and pl.get("SQ_THREAD_CYCLES_VALU") is not None:   # lanes that work per vector instruction (64 = all): SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU
                out["active_lanes_per_valu_inst"] = pl["SQ_THREAD_CYCLES_VALU"] / pl["SQ_ACTIVE_INST_VALU"]
                out["lane_utilisation"] = out["active_lanes_per_valu_inst"] / 64.0
            if pl.get("SQ_INSTS_SALU") and pl.get("SQ_INSTS_VALU"):
                out["salu_per_valu_inst"] = pl["SQ_INSTS_SALU"] / pl["SQ_INSTS_VALU"]
            return out
    return None


def layout_variants():
    """Other picture layouts of the same search, quoted from the newest committed bench lines under profiles/ (NOT measured in this run): the reference cfg's own layout -
    one tile per picture - with WaveFrontSynchro rows as streams, and the plain one-tile-per-picture stream."""
    import glob
    out = {}
    for key, pat in (("tiles_1x1_wpp_1080p", "r*_bench_1080p_tiles_1x1_wpp*.json"), ("tiles_1x1_wpp_4k10", "r*_bench_4k10_tiles_1x1_wpp*.json"), ("tiles_4x2_1080p", "r*_bench_tiles_4x2*.json")):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        for f in reversed(files):
            try:
                d = json.loads([l for l in open(f) if l.startswith("{")][-1])
            except Exception:
                continue
            out[key] = {"value": d["value"], "unit": d["unit"], "workload": d["config"]["workload"], "tiling": d["config"].get("tiling"), "source": os.path.relpath(f, ROOT), "measured_here": False}
            break
    return out


def _cpu_job(job):
    """one process of the frame-parallel CPU baseline: the oracle on the sample crop (module level: spawned workers import it)"""
    crop, sw, sh, sp, kw, classifier = job
    import oracle_lib as O
    forest = None
    if classifier:
        pkg = importlib.import_module("reduce-complexity-for-intra-coding-of-vvc_amd")
        forest = pkg.load_forest(os.path.join(ROOT, "reduce-complexity-for-intra-coding-of-vvc_amd", "forests", "partition_qp32.npz"))
    O.compress_frame(crop, sw, sh, sp, forest=forest, **kw)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--bit-depth", type=int, default=8, choices=(8, 10), help="10: uint16 planes (BASELINE configs 4 and 5), algorithmic bytes double")
    ap.add_argument("--frames", type=str, default="auto",
                    help="frames per step and rank; auto = enough frames for ~4 full waves of resident CTU streams, fewer for long runs (see RUN_BUDGET_S)")
    ap.add_argument("--tiles", type=str, default="auto",
                    help="CxR uniform tile grid (1x1 = the reference cfg's single tile: one stream per frame; 4x2; ...); auto = one tile per CTU")
    ap.add_argument("--lib", type=str, default=None, help="alternative build of the HIP library (experiments only)")
    ap.add_argument("--tools", type=lambda v: int(v, 0), default=0xfff,
                    help="VVCX_TOOL_* bits; default 0xfff = every tool of BIN/encoder_intra.cfg that reaches the path: MRL | MIP | ISP | LFNST | MTS | TransformSkip | DepQuant | "
                         "RDOQ(TS) | CCLM | JointCbCr | LMCS | CU reuse (0xb5b = the round-2 tool set, for the same-tools delta)")
    ap.add_argument("--lmcs", type=str, default="analysis", choices=("analysis", "model"),
                    help="analysis: what the reference's picture analysis decides for this content: LMCS off for every 8-bit picture and for full-range 10-bit ones "
                         "(EL/EncReshape.cpp, see DESIGN.md); model: the slice carries a typical SDR model (limited-range luma), to time chroma residual scaling")
    ap.add_argument("--wpp", action="store_true", help="cfg WaveFrontSynchro 1 (VVCX_TOOL_WPP): the CTU rows of a tile as streams one CTU behind the row above; with --tiles 1x1 this is "
                    "the standard's own parallelism for the reference's one-tile pictures (its bitstream differs from the default cfg's)")
    ap.add_argument("--classifier", action="store_true",
                    help="BASELINE config 3 flavour: the fork's FAST_ALGORITHM with the shipped forest (forests/partition_qp32.npz) on the device")
    ap.add_argument("--chroma-texture", type=float, default=0.5,
                    help="fraction of the luma texture mixed into the synthetic chroma planes (0 = smooth chroma)")
    ap.add_argument("--emit-payload", action="store_true", help="run the slice_data writer in the timed steps also at N = 1 (N > 1 always does: the job ends with the bitstream gather)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-ctus", type=int, default=8)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the frame-parallel CPU baseline (0 = the host's usable cores, at most 16)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # self-launch: one process per GPU, started before this process initialises anything on the GPU (no exec of an initialised process)
        import subprocess
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()

    pkg = importlib.import_module("reduce-complexity-for-intra-coding-of-vvc_amd")
    W, H = args.width, args.height
    ctus_w, ctus_h = (W + 127) // 128, (H + 127) // 128
    if args.tiles == "auto":
        tc, tr = ctus_w, ctus_h
    else:
        tc, tr = map(int, args.tiles.lower().split("x"))
    if args.wpp:
        args.tools |= pkg.TOOL_WPP
    sp = pkg.slice_params(args.qp, bit_depth=args.bit_depth, dep_quant=bool(args.tools & 0x40))
    bd = args.bit_depth
    b_ctu = B_CTU_8BIT * (2 if bd == 10 else 1)
    forest = None
    if args.classifier:
        args.tools |= pkg.TOOL_FAST
        forest = pkg.load_forest(os.path.join(ROOT, "reduce-complexity-for-intra-coding-of-vvc_amd", "forests", "partition_qp32.npz"))
    if args.frames == "auto":
        probe = pkg.VvcxEncoder(W, H, bd, tile_cols=tc, tile_rows=tr, device=dev, lib_path=args.lib, tools=args.tools, forest=forest)
        streams_per_frame = tc * tr if not args.wpp else sum(len(set(ry for ry in range(ctus_h) if max(i for i in range(tr) if ry >= (i * ctus_h) // tr) == t)) for t in range(tr)) * tc
        args.frames = max(1, (4 * probe.resident_streams()) // streams_per_frame)
        probe.close()
        # A long run (the round-end driver times 20 steps after 5 of warm-up inside a 600 s limit) gets a smaller batch so that all its steps fit
        # RUN_BUDGET_S at a conservative rate; never below two full waves of resident streams.  The batch is part of config.workload.
        fit = int(RUN_BUDGET_S * BUDGET_CTUS_PER_S / ((args.steps + args.warmup) * ctus_w * ctus_h))
        args.frames = max(min(args.frames, fit), max(1, args.frames // 2))
    else:
        args.frames = int(args.frames)
    emit = world > 1 or args.emit_payload           # the N-GPU job ends with the bitstream gather: its ranks run the slice_data writer too
    enc = pkg.VvcxEncoder(W, H, bd, tile_cols=tc, tile_rows=tr, chroma=True, max_frames=args.frames, device=dev, lib_path=args.lib, tools=args.tools, forest=forest, emit_payload=emit)
    lmcs = None
    if args.lmcs == "model" and (args.tools & 0x400):
        # what EncReshape::preAnalyzerLMCS typically signals for SDR limited-range content (tests/golden/lmcs.npz): bins 1..14, a few code words moved to the middle
        d = 6 if bd == 10 else 1
        lmcs = dict(enable=1, chroma_adj=1, min_bin=1, max_bin=14, delta_cw=[0] + [d] * 7 + [d + (2 if bd == 10 else 0)] * 2 + [d] * 5 + [0])
        sp["lmcs"] = lmcs
    elif args.tools & 0x400:
        # the reference's order of things: the picture analysis (vvcx_lmcs_analyze = EncReshape::preAnalyzerLMCS, run on the first picture of the batch) decides whether the
        # slice uses LMCS; for 8-bit and for full-range 10-bit content it says no, like the reference
        first = next(iter(pkg.frames_of_rank(args.frames * world, rank, world)))
        m = pkg.vvcx.lmcs_analyze(pkg.synth_frame(W, H, first, bd, 1000 + first, chroma_texture=args.chroma_texture), bd, args.qp, lib_path=args.lib)
        if m["enable"]:
            lmcs = m; sp["lmcs"] = lmcs
    enc.set_slice(sp["qp"], sp["qp_c"], sp["lam"], sp["dist_weight"], lmcs=lmcs)
    frames = []
    for poc in pkg.frames_of_rank(args.frames * world, rank, world):      # weak scaling: args.frames per rank
        planes = pkg.synth_frame(W, H, poc, bd, 1000 + poc, chroma_texture=args.chroma_texture, limited=(args.lmcs == "model" and lmcs is not None))
        org = [torch.from_numpy(p if bd == 8 else p.view(np.int16)).cuda() for p in planes]      # 16-bit containers: same bits, a dtype torch can hold
        rec = [torch.zeros_like(t) for t in org]
        frames.append((org, rec))
    bind = [([t.data_ptr() for t in o], [t.data_ptr() for t in r], [t.shape[1] for t in o]) for o, r in frames]
    ctus_per_step = args.frames * ctus_w * ctus_h

    nstep = [0]

    def step():
        enc.bind_frames(bind)                      # resets every stream to its tile start; planes stay in HBM
        res = enc.compress_bound_frames()
        nstep[0] += 1
        if rank == 0:                              # a progress line per step on stderr (stdout carries the one JSON line): a run of several minutes must not look hung
            print("[bench] step %d of %d (%d warm-up): %.1f s kernel, %d CTUs" % (nstep[0], args.steps + args.warmup, args.warmup, enc.last_kernel_ms() / 1e3, ctus_per_step), file=sys.stderr, flush=True)
        return res, enc.last_kernel_ms()

    elapsed, outs = pkg.timed_steps(step, args.steps, args.warmup, world, device_sync=torch.cuda.synchronize, device="cuda")
    kernel_ms = [ms for _, ms in outs]
    counters = enc.counters()
    gather = None
    if world > 1:
        # the final bitstream gather (north_star: RCCL over xGMI only here): payloads of this rank's frames -> rank 0, timed on its own
        mine = list(pkg.frames_of_rank(args.frames * world, rank, world))
        local = {poc: [enc.get_payload(i, t) for t in range(tc * tr)] for i, poc in enumerate(mine)}
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        merged = pkg.gather_payloads(local, world, device="cuda")
        torch.cuda.synchronize(); dist.barrier()
        gms = 1e3 * pkg.max_over_ranks(time.perf_counter() - t0, world, "cuda")
        if rank == 0:
            gather = {"ms": gms, "bytes": int(sum(len(b) for tiles in merged.values() for b in tiles)), "frames": len(merged), "ranks_seen": world,
                      "collectives": "all_gather(sizes) + gather(padded) for the index and for the bytes (RCCL)"}

    if rank == 0:
        total_ctus = ctus_per_step * args.steps * world
        value = total_ctus / elapsed
        avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3
        achieved = ctus_per_step * b_ctu / avg_kernel_s / 1e9
        workload = ("%dx%d %d-bit 4:2:0 All-Intra QP%d full RDO, tools 0x%x%s, chroma texture %.2f, %d frame(s)/step/GPU, %dx%d uniform tiles = %d CTU streams per frame"
                    % (W, H, bd, args.qp, args.tools, (" (LMCS model on, limited-range luma)" if args.lmcs == "model" else " (LMCS model chosen by the picture analysis)") if lmcs else "", args.chroma_texture, args.frames, tc, tr, tc * tr))
        traffic, traffic_src, traffic_split = pmc_traffic(workload)
        valu = pmc_valu(workload)
        out = {
            "metric": "CTUs/sec (All-Intra, QP32)", "value": value, "unit": "CTU/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16/int32 samples+coefficients, fp64 RD cost", "data": "synthetic",
            "config": {"workload": workload,
                       "tools": "67 intra modes + PDPC" + (" + MRL" if args.tools & 1 else "") + (" + MIP search (FastMIP 1)" if args.tools & 2 else "") + (" + CCLM (LM, MDLM_L, MDLM_T)" if args.tools & 0x100 else "")
                                + (", DCT-II + explicit MTS (DST-VII/DCT-VIII" + (", as CU-level passes" if args.tools & 8 else ", MTSIntraMaxCand 3") + ")" if args.tools & 0x10 else ", DCT-II")
                                + (" + LFNST (lfnstIdx passes, FastLFNST 1)" if args.tools & 8 else "")
                                + (", dependent quantisation (DepQuant 1)" if args.tools & 0x40 else ", plain quant") + ", dual tree" + (", CU-result reuse (REUSE_CU_RESULTS)" if args.tools & 0x800 else "")
                                + (", FAST_ALGORITHM partition classifier (shipped forest)" if args.tools & 0x1000 else "")
                                + (", JointCbCr" if args.tools & 0x200 else "")
                                + (", ISP (ISPFast 1)" if args.tools & 4 else "") + (", transform skip + RDOQ-TS (TransformSkipFast 1, log2 max size 5)" if args.tools & 0x20 else "")
                                + ((", LMCS: slice model on" if lmcs else ", LMCS: tool on, switched off for these pictures by the reference's own analysis (8-bit / full-range content, EL/EncReshape.cpp)") if args.tools & 0x400 else "")
                                + "; off: " + ", ".join(n for n, b in (("ISP", 4), ("transform skip", 0x20), ("LMCS", 0x400)) if not args.tools & b) + ("-" if (args.tools & 0x424) == 0x424 else "") + " (BDPCM is off in the cfg)"
                                + ("" if args.tools & 8 else ", LFNST off") + ("" if args.tools & 0x40 else ", DepQuant off") + ("" if args.tools & 0x200 else ", JointCbCr off")
                                + "; leaf operators, syntax and reconstruction are pinned to the reference (CommonLib + decoder + EncReshape), the search decisions (EncCu / EncModeCtrl / IntraSearch restatement) are pinned only through the decoder accepting and reconstructing the streams",
                       "tiling": ("one tile per CTU (every CTU an independent stream; the reference's cfg codes one tile per picture: against that layout this tiling costs "
                                  "+11.3 % BD-rate on luma / +20.7 % on YUV at 1080p, WPP +0.2 / +1.3 %, 4x2 tiles +2.2 / +2.7 %: profiles/r04_layout_table.json)" if (tc, tr) == (ctus_w, ctus_h)
                                  else "%dx%d uniform tiles" % (tc, tr)) + (", WaveFrontSynchro 1: every CTU row of a tile is a stream that runs one CTU behind the row above" if args.wpp else ""),
                       "ctus_per_step": ctus_per_step, "parallelism": ("CTU rows as streams that migrate between the workgroups of the resident slots (a row is taken while its next CTU is ready and put back), frames sharded over ranks"
                                       if args.wpp else "1 workgroup per CTU stream over a work queue of resident slots, frames sharded over ranks")},
            "roofline": {"bound": "hbm", "limiter": "not HBM: VALU issue and the serial chains of one CTU stream (mode controller, trellis, CABAC estimator); see valu.issue_frac and DESIGN.md", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_split": traffic_split, "traffic_source": traffic_src, "kernel": ("vvcx_compress_wpp_kernel_" if args.wpp else "vvcx_compress_kernel_") + ("u8" if bd == 8 else "u16"), "kernel_ms": 1e3 * avg_kernel_s,
                         "algorithmic_bytes_per_launch": ctus_per_step * b_ctu, "valu": valu},
            "variants": layout_variants(),
            "emit_payload": bool(emit),
            "work": {"satd_candidates_per_launch": int(counters[0]), "rd_tu_evaluations_per_launch": int(counters[1]),
                     "rd_pixels_per_launch": int(counters[2]), "nodes_per_launch": int(counters[3]),
                     "rd_pixels_per_s": float(counters[2]) / avg_kernel_s, "satd_candidates_per_s": float(counters[0]) / avg_kernel_s,
                     "rd_tu_evaluations_per_s": float(counters[1]) / avg_kernel_s},
        }
        # SURVEY 8d (iii), the diagnostic traffic model: what an implementation that staged nothing in LDS would pull per launch - every full-RD TU evaluation its original
        # samples and reference lines, every winner written and read once - from the work counters (TU shapes approximated by squares of the mean TU area)
        if counters[1]:
            side = (float(counters[2]) / float(counters[1])) ** 0.5
            bps = 2 if bd == 10 else 1
            out["work"]["diagnostic_traffic_model_bytes_per_launch"] = float(counters[2]) * bps + float(counters[1]) * (4 * side + 1) * 2 + ctus_per_step * b_ctu

        # secondary kernel (not part of the timed step): in-loop deblocking of the pictures just coded, an HBM-bound pass
        try:
            if lmcs:
                enc.lmcs_inverse_reco()                 # the loop filters work in the original domain
            db_ms = min(enc.deblock_bound_frames() for _ in range(1))
            db_bytes = args.frames * (W * H * 3 // 2) * (2 if bd == 10 else 1) * 2          # every sample read once and written once
            out["deblock"] = {"kernel": "vvcx_deblock_kernel_u8" if bd == 8 else "vvcx_deblock_kernel_u16", "launches": 2, "ms": db_ms, "frames": args.frames,
                              "algorithmic_bytes": db_bytes, "achieved_GBps": db_bytes / (db_ms / 1e3) / 1e9 if db_ms > 0 else None, "frac_of_hbm_peak": db_bytes / (db_ms / 1e3) / 1e9 / HBM_PEAK_GBS if db_ms > 0 else None,
                              "note": "deblocks the pictures of the last step once (a second call would filter filtered samples)"}
        except Exception as ex:                                   # an older --lib build without the kernel
            out["deblock"] = {"error": str(ex)}
        # the encoder's SAO statistics on the deblocked pictures (the O(samples) half of the SAO parameter decision): original and reconstruction read once
        try:
            if not lmcs:
                sao_stats, st_ms = enc.sao_statistics_bound_frames(1)
                st_bytes = args.frames * (W * H * 3 // 2) * (2 if bd == 10 else 1) * 2          # original + deblocked samples read once
                out["sao_statistics"] = {"kernel": "vvcx_sao_stats_kernel_%s" % ("u8" if bd == 8 else "u16"), "launches": 1, "ms": st_ms, "frames": args.frames, "algorithmic_bytes": st_bytes,
                                         "achieved_GBps": st_bytes / (st_ms / 1e3) / 1e9 if st_ms > 0 else None, "frac_of_hbm_peak": st_bytes / (st_ms / 1e3) / 1e9 / HBM_PEAK_GBS if st_ms > 0 else None}
                # the RD half of the decision: host code of the library, serial over the CTUs of a picture (the SAO context models travel from CTU to CTU)
                t0 = time.time()
                lam3 = [sp["lam"], sp["lam"] / sp["dist_weight"][0], sp["lam"] / sp["dist_weight"][1]]
                sao_decided = np.stack([pkg.vvcx.sao_decide(sao_stats[f], W, H, bd, lam3, sp["qp"], tc, tr, lib_path=args.lib) for f in range(args.frames)])
                out["sao_statistics"]["decision_host_ms_per_picture"] = (time.time() - t0) * 1e3 / args.frames
                out["sao_statistics"]["decided_modes_off_new_merge"] = [int(v) for v in np.bincount(sao_decided[:, :, :, 0].ravel(), minlength=3)]
        except Exception as ex:
            out["sao_statistics"] = {"error": str(ex)}
        # third kernel pair (not part of the timed step): sample adaptive offset with seeded per-CTU parameters on the deblocked pictures - an HBM-bound pass
        # (every sample read once and written once; the copy the filter reads its unfiltered neighbours from doubles the traffic)
        try:
            decided = "sao_statistics" in out and "decision_host_ms_per_picture" in out["sao_statistics"]
            prm = sao_decided if decided else np.stack([pkg.sao_test_params(900 + i, W, H, tc, tr) for i in range(args.frames)])
            sao_ms = enc.sao_bound_frames(prm, lf_across_tiles=1)
            sao_bytes = args.frames * (W * H * 3 // 2) * (2 if bd == 10 else 1) * 2
            out["sao"] = {"kernel": "vvcx_sao_copy_kernel + vvcx_sao_kernel (%s)" % ("u8" if bd == 8 else "u16"), "launches": 2, "ms": sao_ms, "frames": args.frames, "algorithmic_bytes": sao_bytes,
                          "achieved_GBps": sao_bytes / (sao_ms / 1e3) / 1e9 if sao_ms > 0 else None, "frac_of_hbm_peak": sao_bytes / (sao_ms / 1e3) / 1e9 / HBM_PEAK_GBS if sao_ms > 0 else None,
                          "note": ("parameters decided by the library from the statistics above (vvcx_sao_decide)" if decided else
                                   "parameters are seeded test values (synth.sao_test_params: about 80 % of the CTUs filtered)")}
        except Exception as ex:
            out["sao"] = {"error": str(ex)}
        # fourth kernel pair (not part of the timed step): the adaptive loop filter with seeded parameter sets and per-CTU choices on the pictures SAO left - an HBM-bound
        # stencil pass (every sample read once and written once; the taps and the classifier's window come from an LDS tile)
        try:
            base = pkg.alf_test_params(950, W, H)
            nalt = int(base["aps"][base["chroma_aps"], 627])
            prms = [dict(base, ctu=pkg.alf_test_params(951 + i, W, H)["ctu"] % np.array([2, 2, 2, 16 + len(base["luma_aps"]), nalt, nalt])) for i in range(args.frames)]
            alf_ms = enc.alf_bound_frames(prms)
            alf_bytes = args.frames * (W * H * 3 // 2) * (2 if bd == 10 else 1) * 2
            out["alf"] = {"kernel": "vvcx_alf_copy_kernel + vvcx_alf_kernel (%s)" % ("u8" if bd == 8 else "u16"), "launches": 2, "ms": alf_ms, "frames": args.frames, "algorithmic_bytes": alf_bytes,
                          "achieved_GBps": alf_bytes / (alf_ms / 1e3) / 1e9 if alf_ms > 0 else None, "frac_of_hbm_peak": alf_bytes / (alf_ms / 1e3) / 1e9 / HBM_PEAK_GBS if alf_ms > 0 else None,
                          "note": "parameter sets and per-CTU choices are seeded test values (synth.alf_test_params: about 80 % of the CTUs filtered, fixed and signalled filter sets); the parameter decision is not part of the library"}
        except Exception as ex:
            out["alf"] = {"error": str(ex)}
        if not args.no_cpu_baseline and world == 1:
            print("[bench] timed region done (%.1f s for %d steps); CPU baseline sample ..." % (elapsed, args.steps), file=sys.stderr, flush=True)
            import oracle_lib as O
            n = max(1, min(args.cpu_sample_ctus, ctus_w * ctus_h))
            cw = min(ctus_w, 4)
            chh = max(1, n // cw)
            sw, sh = min(W, cw * 128), min(H, chh * 128)
            planes = pkg.synth_frame(W, H, 0, bd, 1000, chroma_texture=args.chroma_texture, limited=(args.lmcs == "model" and lmcs is not None))
            crop = [planes[0][:sh, :sw], planes[1][:sh // 2, :sw // 2], planes[2][:sh // 2, :sw // 2]]
            t1 = time.perf_counter()
            O.compress_frame(crop, sw, sh, sp, tile_cols=(sw + 127) // 128, tile_rows=(sh + 127) // 128, tools=args.tools, forest=forest, bit_depth=bd)
            dt = time.perf_counter() - t1
            nct = ((sw + 127) // 128) * ((sh + 127) // 128)
            out["cpu_baseline"] = {"value": nct / dt, "unit": "CTU/s", "cores": 1, "kind": "port",
                                   "sample": "top-left %dx%d crop (%d CTUs, one tile per CTU) of the same frame, same QP/tools, oracle/ built -O2 -mavx2, %.1f s" % (sw, sh, nct, dt),
                                   "reference_encoder": {"value": 0.44, "unit": "CTU/s", "cores": 1, "measured_here": False,
                                                         "note": "QUOTED, not measured in this run: the reference's own EncoderApp (full encoder_intra.cfg, AVX2) on 416x240 QP32, measured once in the authoring container (BASELINE.md); its build needs OpenCV, which this image lacks"}}
            # the frame-parallel CPU figure beside the one-core one: N processes, each coding the same sample (frames of an All-Intra sequence
            # are independent, so N encoder processes on N frames is how the reference scales on a host)
            try:
                import multiprocessing as mp
                nproc = args.cpu_procs or max(1, min(16, len(os.sched_getaffinity(0))))
                if nproc > 1:
                    job = (crop, sw, sh, sp, dict(tile_cols=(sw + 127) // 128, tile_rows=(sh + 127) // 128, tools=args.tools, bit_depth=bd), args.classifier)
                    t2 = time.perf_counter()
                    with mp.get_context("spawn").Pool(nproc) as pool:      # spawn: the parent has initialised the GPU runtime
                        pool.map(_cpu_job, [job] * nproc)
                    dtn = time.perf_counter() - t2
                    out["cpu_baseline"]["frame_parallel"] = {"value": nproc * nct / dtn, "unit": "CTU/s", "cores": nproc, "kind": "port",
                                                             "sample": "%d processes x the same %d-CTU sample, %.1f s wall incl. process start" % (nproc, nct, dtn)}
            except Exception as ex:
                out["cpu_baseline"]["frame_parallel"] = {"error": str(ex)}
        if gather:
            out["gather"] = gather
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
