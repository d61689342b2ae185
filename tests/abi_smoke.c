/* A plain-C translation unit that includes include/jarvis_hip.h: proves the header is
 * valid C (not only C++), pins the layout of jh_predictor_config that
 * jarvis-hybridnet_amd/_native.py mirrors by hand, and -- linked against
 * libjarvis_hip.so -- calls the entry points that need no GPU.  Built and run by
 * tests/test_native_abi.py::test_header_is_valid_c_and_links (gcc, no hipcc). */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "jarvis_hip.h"

_Static_assert(sizeof(jh_predictor_config) == 84, "jh_predictor_config layout (mirrored in _native.py)");
_Static_assert(offsetof(jh_predictor_config, roi_cube_size) == 16, "roi_cube_size offset");
_Static_assert(offsetof(jh_predictor_config, time_batch) == 40, "time_batch offset");
_Static_assert(offsetof(jh_predictor_config, mean) == 56, "mean offset");
_Static_assert(offsetof(jh_predictor_config, std) == 68, "std offset");
_Static_assert(offsetof(jh_predictor_config, precision) == 80, "precision offset");

int main(void) {
  jh_params* p = NULL;
  float w[4] = {1.f, 2.f, 3.f, 4.f};
  if (jh_abi_version() != JH_ABI_VERSION) {
    fprintf(stderr, "ABI version mismatch: header %d, library %d\n", JH_ABI_VERSION, jh_abi_version());
    return 1;
  }
  if (jh_params_create(&p) != 0 || jh_params_set(p, "weights_cat", w, 3) != 0) return 2;
  /* a failing call must set the thread's error text (bad argument: NULL key) */
  if (jh_params_set(p, NULL, w, 3) == 0 || strlen(jh_last_error()) == 0) return 3;
  jh_params_destroy(p);
  /* workspace sizes are pure host arithmetic */
  if (jh_reproject_workspace_bytes(12, 23, 130, 64) <= 0) return 4;
  if (jh_softargmax_workspace_bytes(1, 23, 32) <= 0) return 5;
  if (jh_reconstruct_workspace_bytes(12) <= 0) return 6;
  printf("abi %d config %zu\n", jh_abi_version(), sizeof(jh_predictor_config));
  return 0;
}
