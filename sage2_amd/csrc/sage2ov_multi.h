// sage2_amd/csrc/sage2ov_multi.h -- multi-GPU driver of steps 2-3 in C++ (see sage2ov_multi.cpp): used by the sage2ov CLI (`--gpus G`).
#pragma once
#include <string>
#include <vector>
#include "sage2ov.h"

namespace sage2ov_multi {
// ctx[r]: a context created with rank r of world ctx.size(), holding the organised read set (rank 0 organises, the others import its image:
// sage2ov_reads_export_words / _import_words); devices[r]: its HIP device.  share_gpu = false: RCCL collectives, one distinct GPU per rank;
// true: all ranks on one device, exchanges by device copies (rehearsal on a box with fewer GPUs than ranks).  On return every context holds
// the complete canonical edge list (sage2ov_overlap_convert has run).  Returns a SAGE2OV_* status; err names the failing rank.  A rank that fails
// (library error, allocation, collective) takes the others down with it instead of leaving them inside a collective: every rank returns and the call
// reports the first failure.  fail_rank >= 0 (tests only, `sage2ov --fail-rank r`): that rank fails before its first step.
int run_steps23(const std::vector<sage2ov_ctx*>& ctx, const std::vector<int>& devices, bool share_gpu, std::string& err, int fail_rank = -1);
}
