// Sustained MFMA rate under the package power limit: v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16, operands in
// registers (no memory traffic), 2 waves / SIMD on every CU.  Which tile shape delivers more FLOP/s per watt decides
// whether re-tiling the network kernel around 32x32 tiles could pay.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void burn(float* out, int iters, float seed) {
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 8; ++k) {
            a[i][k] = (_Float16)(seed + 0.001f * (float)((threadIdx.x * 7 + i * 3 + k) & 63));
            b[i][k] = (_Float16)(seed - 0.002f * (float)((threadIdx.x * 5 + i + k * 11) & 63));
        }
    float acc_out = 0.f;
    if (MODE == 0) {
        f4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) acc_out += c[i][0] + c[i][3];
    } else {
        f16v c[4];
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 16; ++k) c[i][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[(i + 2 * r) & 3], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) acc_out += c[i][0] + c[i][15];
    }
    if (acc_out == 12345.678f) out[0] = acc_out;
}

template <int MODE>
double run(float* out, double seconds, double flop_per_iter_per_wave) {
    const int blocks = 256, threads = 512, iters = 20000;
    hipLaunchKernelGGL(burn<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, 0.5f);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    while (el < seconds) {
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(burn<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5f);
        hipDeviceSynchronize();
        launches += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double flop = (double)launches * blocks * (threads / 64) * iters * flop_per_iter_per_wave;
    return flop / el / 1e12;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    float* out;
    hipMalloc(&out, 64);
    // per iteration and wave: MODE 0: 16 MFMAs x 16*16*32*2 FLOP; MODE 1: 8 MFMAs x 32*32*16*2 FLOP
    for (int rep = 0; rep < 2; ++rep) {
        const double t0 = run<0>(out, secs, 16.0 * 16 * 16 * 32 * 2);
        printf("v_mfma_f32_16x16x32_f16: %.0f TFLOP/s sustained over %.0f s\n", t0, secs);
        fflush(stdout);
        const double t1 = run<1>(out, secs, 8.0 * 32 * 32 * 16 * 2);
        printf("v_mfma_f32_32x32x16_f16: %.0f TFLOP/s sustained over %.0f s\n", t1, secs);
        fflush(stdout);
    }
    return 0;
}
