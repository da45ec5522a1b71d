// temporary: read-level entry points until pipeline.hip lands
#include "thermite_internal.h"
extern "C" {
int32_t thm_align_batch(thm_aligner*, const uint8_t*, const uint64_t*, uint64_t, thm_batch_view*) { return THM_ERR_INTERNAL; }
int32_t thm_batch_upload(thm_aligner*, const uint8_t*, const uint64_t*, uint64_t) { return THM_ERR_INTERNAL; }
int32_t thm_batch_run(thm_aligner*) { return THM_ERR_INTERNAL; }
int32_t thm_batch_sync(thm_aligner*) { return THM_ERR_INTERNAL; }
int32_t thm_batch_fetch(thm_aligner*, thm_batch_view*) { return THM_ERR_INTERNAL; }
int32_t thm_smems_batch(thm_aligner*, const uint8_t*, const uint64_t*, uint64_t, uint64_t, thm_mems_view*) { return THM_ERR_INTERNAL; }
}
