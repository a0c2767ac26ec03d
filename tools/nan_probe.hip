// Prints the bit patterns the GPU produces for invalid fp32 operations (used to decide how NaNs order as integers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void probe(const float *in, unsigned *out)
{
    float inf = in[0], zero = in[1], one = in[2], ninf = in[3];
    out[0] = __float_as_uint(inf - inf);
    out[1] = __float_as_uint(zero * inf);
    out[2] = __float_as_uint(__fmaf_rn(one, inf, ninf));
    out[3] = __float_as_uint(__fmaf_rn(zero, inf, one));
    float n = zero * inf;
    out[4] = __float_as_uint(-n);
    out[5] = __float_as_uint(__fmaf_rn(one, one, -n));
    out[6] = __float_as_uint(fminf(n, -n));
    out[7] = __float_as_uint(__fmaf_rn(-one, inf, inf));
    out[8] = __float_as_uint(zero / zero);
    out[9] = __float_as_uint(sqrtf(-one));
    out[10] = __float_as_uint(fmaxf(inf - inf, one));
    out[11] = __float_as_uint(__fmaf_rn(one, ninf, inf));
}
int main()
{
    float h[4] = {__builtin_inff(), 0.0f, 1.0f, -__builtin_inff()};
    float *d; unsigned *o; unsigned r[12];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    probe<<<1, 1>>>(d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    const char *names[12] = {"inf-inf", "0*inf", "fma(1,inf,-inf)", "fma(0,inf,1)", "-(0*inf)", "fma(1,1,-nan)", "fmin(nan,-nan)",
                             "fma(-1,inf,inf)", "0/0", "sqrt(-1)", "fmax(nan,1)", "fma(1,-inf,inf)"};
    for (int i = 0; i < 12; i++) printf("%-18s %08x\n", names[i], r[i]);
    return 0;
}
