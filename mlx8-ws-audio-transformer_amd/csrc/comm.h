// awt_comm: RCCL communicator + side stream used by the fine-tune step's gradient exchange (comm.hip).
#pragma once
#include <string.h>
#include "common.h"

struct awt_comm {
  awt_ctx* ctx = nullptr;
  void* nccl = nullptr;          // ncclComm_t
  int rank = 0, world = 1;
  hipStream_t side = nullptr;    // reductions of finished layer groups run here, next to the remaining backward
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // [0..3] producer -> side (round robin), [4] side -> consumer
  int next_ev = 0;
  // timing of the in-backward bucket reductions (awt_comm_bucket_stats): one event pair per comm_reduce_async on the side stream since the last read-out
  struct Span { hipEvent_t a = nullptr, b = nullptr; size_t bytes = 0; };
  Span spans[8];
  int n_spans = 0;
  bool timing = false;
};

int comm_reduce_async(awt_comm* m, float* buf, size_t n, hipStream_t producer);
int comm_join(awt_comm* m, hipStream_t consumer);
