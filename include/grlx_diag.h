/* grlx_diag.h -- diagnostic exports of libgrlx.so: not part of the drop-in boundary (include/grlx.h), kept stable only for the
 * repository's own tools, tests and bench.py.  Every symbol the library exports is declared in one of the two headers
 * (tests/test_capi_symbols.py checks exported == declared). */
#ifndef GRLX_DIAG_H_
#define GRLX_DIAG_H_

#include "grlx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The environment server of the pendulum rollout kernels (grl_amd/csrc/grlx_env_server.h): in the last launch of the context that had
 * it, how many replicas took every environment step from it and how many gave up waiting and integrated themselves; both 0 when no
 * launch of this context had it. */
int grlx_env_server_counts(grlx_ctx *ctx, int *served, int *fell_back);
/* The raw mailboxes of the environment server after the last launch (1 KB per replica; GRLX_ENV_SERVER_STATS builds leave cycle counts
 * in them): at most `bytes` bytes to `out`. */
int grlx_env_server_debug(grlx_ctx *ctx, void *out, size_t bytes);
/* Shader-clock stamps of the LAST fqi_epochs_kernel launch (GRLX_FQI_STAMPS=1 at grlx_fqi_create): count = n_replicas * 16 * 4 * 8. */
int grlx_fqi_debug_stamps(grlx_fqi_ctx *ctx, unsigned long long *out, int count);

#ifdef __cplusplus
}
#endif
#endif /* GRLX_DIAG_H_ */
