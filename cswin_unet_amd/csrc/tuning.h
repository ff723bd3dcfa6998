// Tuning aids: environment variables that select between measured alternatives (profiles/round1_notes.md,
// profiles/round2_notes.md; the list with meanings is DESIGN.md section 8).  They are NOT configuration: the defaults are the
// measured choices and nothing in the package sets them.  Read ONCE per process into one immutable structure -- entry points
// consult the structure, never the environment, and there is no other process-wide state behind them.
#pragma once
#include <stdio.h>
#include <stdlib.h>

struct CswinTuning {
    // tiled GEMM family (gemm.hip)
    int gemm_tile, gemm_kw, gemm_pad_lds, gemm_split_wgs, gemm_batch_wgs, gemm_batch_even, gemm_tail_merge;
    double gemm_pen2;
    // bf16 mode
    int wgrad16_on, w16_wgs, w16_even, w16_dma;             // wgrad16.hip
    int gemm16_on, gemm16_stages, gemm16_tm;                // gemm16.hip
    // attention, CARAFE
    int attn_bwd_two_pass, attn_fwd_qsplit, carafe_generic;
};

inline const CswinTuning& cswin_tuning() {
    static const CswinTuning t = [] {
        auto geti = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
        CswinTuning c = {};
        c.gemm_tile = geti("CSWIN_GEMM_TILE", 0);                // 1 = 64x64, 2 = 64x32, 0 = cost model
        c.gemm_kw = geti("CSWIN_GEMM_KW", 0);                    // k-split wave groups 1 / 2 / 4, 0 = heuristic
        c.gemm_pad_lds = geti("CSWIN_GEMM_PAD_LDS", 0);          // extra dynamic LDS (caps residency)
        c.gemm_split_wgs = geti("CSWIN_GEMM_SPLIT_WGS", 768);    // workgroup target of a stand-alone weight-gradient split
        c.gemm_batch_wgs = geti("CSWIN_GEMM_BATCH_WGS", 0);      // per-problem target of the batched weight gradient (0 = 1024 / n)
        c.gemm_batch_even = geti("CSWIN_GEMM_BATCH_EVEN", 1);    // 0 = shares in proportion to the work (slower in fp32)
        c.gemm_tail_merge = geti("CSWIN_GEMM_TAIL_MERGE", 1);    // 0 = qkv data gradient and the weight-gradient batch as two launches
        c.gemm_pen2 = getenv("CSWIN_GEMM_PEN2") ? atof(getenv("CSWIN_GEMM_PEN2")) : 1.20;     // per-flop penalty of the 64x32 tile
        c.wgrad16_on = geti("CSWIN_WGRAD16", 1);                 // 0 = tiled family for bf16 weight gradients
        c.w16_wgs = geti("CSWIN_W16_WGS", 768);
        c.w16_even = geti("CSWIN_W16_EVEN", 0);                  // 1 = equal workgroup share per problem
        c.w16_dma = geti("CSWIN_W16_DMA", 1);                    // 0 = register path for every problem
        c.gemm16_on = geti("CSWIN_GEMM16", 1);                   // 0 = tiled family where the LDS-DMA GEMM would run
        c.gemm16_stages = geti("CSWIN_GEMM16_STAGES", 0);        // 2 .. 4 steps in flight, 0 = default (3)
        c.gemm16_tm = geti("CSWIN_GEMM16_TM", 0);                // 64 / 128 tile rows, 0 = by workgroup count
        c.attn_bwd_two_pass = getenv("CSWIN_ATTN_BWD_TWO_PASS") != nullptr;      // large-window backward for every window size
        c.attn_fwd_qsplit = geti("CSWIN_ATTN_FWD_QSPLIT", 0);    // 1 / 2 query-split workgroups per unit, 0 = heuristic
        c.carafe_generic = getenv("CSWIN_CARAFE_GENERIC") != nullptr;            // disable the fused CARAFE4 backward
        return c;
    }();
    return t;
}
