#!/usr/bin/env python3
"""Per-kernel-family matrix-pipe utilisation from a rocprofv3 --pmc pass (tools/pmc_mfma.sh).
MfmaUtil (gfx94x formula of rocprof's derived counters, which gfx950 falls back to) =
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 4 SIMDs x 32 CUs)   [busy matrix-pipe cycles / available SIMD cycles]
rocprofv3 sums both counters over the 8 XCDs (each XCD has its own GRBM), so the denominator uses the 32 CUs of ONE XCD;
the FLOP-derived fractions of the three texture-field kernels (0.82 / 0.77 / 0.83 of the f32 MFMA peak) agree with it.
SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md, PMC units)."""
import csv, glob, sys, collections

FAM = [("gemm_conv (k_gemm144, k_gemm_pipe, k_gemm8, k_splitk_reduce)", ("k_gemm144", "k_gemm_pipe", "k_gemm8", "k_splitk_reduce")),
       ("attention (k_attention_dma)", ("k_attention",)), ("groupnorm", ("k_gn_",)), ("layernorm", ("k_layernorm",)),
       ("texture field forward (k_uvmlp_fwd)", ("k_uvmlp_fwd",)), ("texture field dgrad (k_uvmlp_dgrad)", ("k_uvmlp_dgrad",)),
       ("texture field wgrad (k_uvmlp_wgrad)", ("k_uvmlp_wgrad",))]
for d in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for t in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(t)):
            for f, keys in FAM:
                if any(k in r['Kernel_Name'] for k in keys):
                    agg[f][r['Counter_Name']] += float(r['Counter_Value']); n[f].add(r['Dispatch_Id'])
                    break
    print(f"== {d}")
    for f, v in agg.items():
        gui = v.get('GRBM_GUI_ACTIVE', 0.0)
        util = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 4 * 32) if gui else float('nan')
        wc = v.get('SQ_WAVE_CYCLES', 0.0) or float('nan')
        print(f"{f:52s} dispatches {len(n[f]):5d}  MfmaUtil {100 * util:5.1f} %   wave cycles: active {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:4.1f} %"
              f" (VALU {100 * v.get('SQ_ACTIVE_INST_VALU', 0) / wc:4.1f} %), waiting {100 * v.get('SQ_WAIT_ANY', 0) / wc:4.1f} %,"
              f" issue-stalled {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:4.1f} %")
