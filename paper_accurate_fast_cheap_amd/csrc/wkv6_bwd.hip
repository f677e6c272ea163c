// WKV-6 backward for gfx950 (placeholder until the chunked sweeps land; see DESIGN.md section "backward").
#include "pafc_common.h"
#include "../../include/pafc_wkv6.h"

extern "C" {

size_t pafc_wkv6_bwd_workspace_bytes(int, int, int, int, int) { return 0; }

int pafc_wkv6_backward(int, int, int, int, int, const void *, const void *, const void *, const void *, const void *,
                       const void *, void *, void *, void *, void *, void *, int, int, void *, size_t, pafc_stream_t) {
    return PAFC_ERR_UNSUPPORTED;
}

}  // extern "C"
