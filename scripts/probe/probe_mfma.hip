// Probe of gfx950 instruction semantics the bf16 decoder kernel relies on (run on the GPU box):
//   1. ds_read_b64_tr_b16 lane/element mapping, with contiguous and with scrambled per-lane addresses
//   2. operand / result layouts of v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void tr_probe(const int* lane_elem_off, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (unsigned short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(s + lane_elem_off[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

// A [M][K], B [K][N] row-major floats holding small integers
__global__ void mfma16_probe(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 15, kg = l >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[i * 32 + 8 * kg + j]; b[j] = (__bf16)B[(8 * kg + j) * 16 + i]; }
    floatx4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * kg + r) * 16 + i] = c[r];      // assumed: row = 4*(l/16)+r, col = l%16
}
__global__ void mfma32_probe(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 31, g = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[i * 16 + 8 * g + j]; b[j] = (__bf16)B[(8 * g + j) * 32 + i]; }
    floatx16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = c[r];
}

template <typename T> T* dev(const std::vector<T>& h) { T* d; hipMalloc(&d, h.size() * sizeof(T)); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d; }

int main() {
    int fails = 0;
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<int> off(64);
        for (int l = 0; l < 64; ++l) off[l] = variant == 0 ? 4 * l : 4 * ((l * 37 + 11) % 1024);      // 8-byte aligned chunks
        std::vector<unsigned short> out(256, 0);
        int* doff = dev(off); unsigned short* dout = dev(out);
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, doff, dout);
        hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int src_lane = 16 * (l / 16) + 4 * j + (l % 16) / 4;
                const int expect = off[src_lane] + (l % 16) % 4;
                if (out[l * 4 + j] != expect) ++bad;
            }
        printf("tr_b16 variant %d: %s (%d mismatches)\n", variant, bad ? "FAIL" : "PASS", bad);
        if (bad) { ++fails; for (int l = 0; l < 64; ++l) printf("  lane %2d off %4d -> %4d %4d %4d %4d\n", l, off[l], out[l*4], out[l*4+1], out[l*4+2], out[l*4+3]); }
    }
    {
        std::vector<float> A(16 * 32), B(32 * 16), D(256, 0), R(256, 0);
        srand(1);
        for (auto& v : A) v = (float)(rand() % 7 - 3);
        for (auto& v : B) v = (float)(rand() % 5 - 2);
        for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + n]; R[i * 16 + n] = s; }
        float *dA = dev(A), *dB = dev(B), *dD = dev(D);
        hipLaunchKernelGGL(mfma16_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        int bad = 0; for (int e = 0; e < 256; ++e) bad += D[e] != R[e];
        printf("mfma 16x16x32 bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    {
        std::vector<float> A(32 * 16), B(16 * 32), D(1024, 0), R(1024, 0);
        for (auto& v : A) v = (float)(rand() % 7 - 3);
        for (auto& v : B) v = (float)(rand() % 5 - 2);
        for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + n]; R[i * 32 + n] = s; }
        float *dA = dev(A), *dB = dev(B), *dD = dev(D);
        hipLaunchKernelGGL(mfma32_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0; for (int e = 0; e < 1024; ++e) bad += D[e] != R[e];
        printf("mfma 32x32x16 bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    return fails;
}
