// Process-wide launch policy of the persistent kernels.
//
// The encoder-sized kernels of a step (panel_gemm.hip, gemm.hip's weight gradients, attn_pool*.hip) run ONE round of workgroups with a
// static share of the work each - one workgroup per CU at 128-160 KiB of LDS (K2: two of 65 KiB).  A CU that somebody else holds for
// the length of a launch (RCCL runs one workgroup per channel while a collective is in flight) cannot take its share: that share
// waits for a free CU and runs as a second round, i.e. the launch takes twice as long whether 4 or 32 CUs are missing (measured
// with a co-running "CU thief", tools/cu_thief.py: 1.41 -> 2.15 ms per step for any N in 4..32, profiles/r05_*_cu_thief.txt).
// The CU budget is the number of CUs these launches size their round for: 256 by default; a data-parallel step that overlaps
// collectives with its backward sets 256 - (reserved CUs), so that the channels find free CUs and the round still fits
// (costs budget/256 of the rate of those kernels, always; reference side: nn.DataParallel, train_MuRCL.py:145).
#include "common.h"

static int g_cu_budget = 256;

extern "C" int murcl_cu_budget(void) { return g_cu_budget; }

// -> the budget in force: `cus` rounded down to a multiple of 8 (one step per XCD), clamped to [64, 256]
extern "C" int murcl_set_cu_budget(int cus) {
    if (cus > 256) cus = 256;
    if (cus < 64) cus = 64;
    g_cu_budget = cus & ~7;
    return g_cu_budget;
}
