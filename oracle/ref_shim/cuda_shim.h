// Host shim that lets g++ compile the reference's CUDA *kernel bodies* unchanged (test infrastructure).
// One "thread" (blockDim = gridDim = 1) runs the kernels' own grid-stride loops over the whole input.
#pragma once
#include <cmath>
#include <cstdint>
#include <type_traits>
#define __global__
#define __device__
#define __host__
#define __inline__ inline
#define __restrict__
struct shim_dim3 { unsigned x, y, z; };
static const shim_dim3 blockDim{1, 1, 1}, blockIdx{0, 0, 0}, threadIdx{0, 0, 0}, gridDim{1, 1, 1};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int3 make_int3(int x, int y, int z) { return {x, y, z}; }
static inline float max(float a, float b) { return a > b ? a : b; }
static inline float min(float a, float b) { return a < b ? a : b; }
template <class T> static inline void atomicAdd(T* p, T v) { *p += v; }
