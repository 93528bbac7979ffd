/* Plain-C consumer of include/gemlite_hip.h: the header must compile as C99 and the library must link and answer the
 * host-only entry points without a GPU.  Built and run by tests/test_host_cpu.py::test_c_consumer_links_and_queries. */
#include <stdio.h>
#include <string.h>

#include "gemlite_hip.h"

int main(void) {
    gemlite_hip_forward_args a;
    memset(&a, 0, sizeof a);
    if (gemlite_hip_abi_version() != GEMLITE_HIP_ABI_VERSION) return 10;
    if (gemlite_hip_query(&a) != GEMLITE_ERR_BAD_ARGUMENT) return 11; /* struct_size 0 */
    a.struct_size = (uint32_t)sizeof a;
    a.matmul_type = GEMLITE_MATMUL_AUTO;
    a.x = a.w_q = a.scales = a.zeros = a.out = (void*)0x1000; /* never dereferenced by query / planning */
    a.M = 1; a.N = 4096; a.K = 4096;
    a.W_nbits = 4; a.group_size = 128; a.unpack_mask = 15; a.elements_per_sample = 8; a.w_pack_bits = 32;
    a.w_dtype = GEMLITE_DT_INT32;
    a.input_dtype = a.output_dtype = a.meta_dtype = a.zeros_dtype = GEMLITE_DT_FP16;
    a.W_group_mode = 4; a.data_contiguous = 1; a.type_id = 104;
    a.stride_xm = 4096; a.stride_xk = 1; a.stride_wk = 4096; a.stride_wn = 1; a.stride_om = 4096; a.stride_on = 1;
    a.stride_meta_g = 4096; a.stride_meta_n = 1;
    if (gemlite_hip_query(&a) != GEMLITE_OK) return 12;
    if (gemlite_hip_workspace_bytes(&a) != 0) return 13; /* the decode GEMV does not split K at this shape */
    printf("%s | %s | %s\n", gemlite_hip_build_info(), gemlite_hip_kernel_name(&a), gemlite_hip_status_string(GEMLITE_ERR_UNSUPPORTED));
    a.channel_scale_mode = 4; /* MX block scales: out of scope */
    return gemlite_hip_query(&a) == GEMLITE_ERR_UNSUPPORTED ? 0 : 14;
}
