// placeholder until the executor lands
#include "common.h"
extern "C" int petr_head_layout(const petr_head_config*, petr_head_layout_t*) { petr_set_error("not built"); return PETR_ERR_UNSUPPORTED; }
extern "C" size_t petr_head_workspace_bytes(const petr_head_config*) { return 0; }
extern "C" int petr_head_fwd(const petr_head_config*, const petr_head_io*, void*) { petr_set_error("not built"); return PETR_ERR_UNSUPPORTED; }
extern "C" int petr_head_bwd_num_stages(const petr_head_config*) { return 0; }
extern "C" int petr_head_bwd_stage_range(const petr_head_config*, int, long*, long*) { return PETR_ERR_UNSUPPORTED; }
extern "C" int petr_head_bwd(const petr_head_config*, const petr_head_io*, const petr_head_grads*, int, int, void*) { return PETR_ERR_UNSUPPORTED; }
extern "C" int petr_head_ws_view(const petr_head_config*, const char*, long*, long*) { return PETR_ERR_UNSUPPORTED; }
