// TEST INFRASTRUCTURE: how far is v_rcp_f32 from the correctly rounded 1 / d?  The filtered pick of k_assign (dsm_math.h,
// pick_seed_fast) takes a pixel's inverse depth from the hardware reciprocal and carries "within one ulp of the correctly
// rounded quotient, denormal quotients flushed to 0" in its error bound.  This walks EVERY float d with (double)d > 0.01 up to
// +inf and prints the largest distance in ulps (normal quotients) and what becomes of the denormal ones.
//     hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tests/cpp/rcp_ulp.hip -o tests/_build/rcp_ulp && tests/_build/rcp_ulp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__global__ void walk(uint32_t first, uint32_t last, unsigned long long *out) {
    unsigned long long max_ulp = 0, denorm_nonzero = 0, denorm_err_max = 0, n = 0;
    for (uint64_t b = (uint64_t)first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= last; b += (uint64_t)gridDim.x * blockDim.x) {
        const float d = __builtin_bit_cast(float, (uint32_t)b);
        const float q = 1.0f / d; // correctly rounded (hipcc's default fp32 divide)
        const float r = __builtin_amdgcn_rcpf(d);
        const uint32_t qb = __builtin_bit_cast(uint32_t, q), rb = __builtin_bit_cast(uint32_t, r);
        n++;
        if ((qb & 0x7f800000u) == 0u) { // denormal (or zero: d = +inf) quotient
            if (rb != 0u) denorm_nonzero++;
            const uint32_t e = rb > qb ? rb - qb : qb - rb; // distance in units of 2^-149
            if (e > denorm_err_max) denorm_err_max = e;
        } else {
            const uint32_t e = rb > qb ? rb - qb : qb - rb;
            if (e > max_ulp) max_ulp = e;
        }
    }
    atomicMax(&out[0], max_ulp);
    atomicAdd(&out[1], denorm_nonzero);
    atomicMax(&out[2], denorm_err_max);
    atomicAdd(&out[3], n);
}

int main() {
    const float t = 0.01f; // (float)0.01 > 0.01: the first float with (double)d > 0.01 is t itself or its lower neighbour
    uint32_t first;
    memcpy(&first, &t, 4);
    first -= 1;
    const uint32_t last = 0x7f800000u; // +inf included: the quotient is 0
    unsigned long long *out, host[4];
    if (hipMalloc(&out, sizeof host) != hipSuccess) return 2;
    hipMemset(out, 0, sizeof host);
    walk<<<4096, 256>>>(first, last, out);
    if (hipMemcpy(host, out, sizeof host, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    printf("{\"floats\": %llu, \"max_ulp_normal\": %llu, \"denormal_quotients_not_flushed\": %llu, \"denormal_err_max_in_2^-149\": %llu}\n", host[3], host[0],
           host[1], host[2]);
    return 0;
}
