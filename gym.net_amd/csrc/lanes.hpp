// lanes.hpp — VEC-wide accesses to one structure-of-arrays row, for every element type the engine streams
// (float / double state, observation, reward; int32 action, counters; uint8 done).
//
// A thread owns VEC consecutive lanes i0 .. i0 + VEC - 1 (i0 a multiple of VEC) and moves them with the widest access the row
// allows: up to 16 bytes per instruction (dwordx4) — 4 floats, 2 doubles, 4 int32 — so a 4-lane float row and a 2-lane double
// row are ONE global_load_dwordx4 each; a VEC-lane row wider than 16 bytes (4 doubles) is split into 16-byte pieces.  The
// launcher (capi: default_policy / apply_policy) only selects VEC > 1 when the row base and stride are aligned for it.
//   NT     non-temporal (streaming) access: `global_load/store ... nt`.  Which streams get it is a measured policy
//          (tools/probe_step.hip): it decides what stays in the 256 MiB Infinity Cache between launches.
//   GUARD  per-element bounds checks; only the last (partial) workgroup of a launch runs the guarded form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gymnet {

template <class T, int N> struct VecOf { typedef T type __attribute__((ext_vector_type(N))); };
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

// elements of T per 16-byte access, capped at VEC
template <class T, int VEC>
constexpr int piece_of() { return (int)(16 / sizeof(T)) < VEC ? (int)(16 / sizeof(T)) : VEC; }

template <class T, int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void load_row(const T *__restrict__ p, int64_t i0, int64_t n, T (&v)[VEC]) {
    if constexpr (VEC > 1 && sizeof(T) > 1) {
        if (!GUARD || i0 + VEC <= n) {
            constexpr int P = piece_of<T, VEC>();
            static_assert(VEC % P == 0, "VEC must be a whole number of 16-byte pieces");
            typedef typename VecOf<T, P>::type V;
#pragma unroll
            for (int q = 0; q < VEC / P; ++q) {
                V t;
                if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const V *>(p + i0 + q * P));
                else t = *reinterpret_cast<const V *>(p + i0 + q * P);
#pragma unroll
                for (int j = 0; j < P; ++j) v[q * P + j] = t[j];
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (!GUARD || i0 + j < n) { if constexpr (NT) v[j] = __builtin_nontemporal_load(p + i0 + j); else v[j] = p[i0 + j]; }
        else v[j] = T(0);
    }
}

template <class T, int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void store_row(T *__restrict__ p, int64_t i0, int64_t n, const T (&v)[VEC]) {
    if constexpr (VEC > 1 && sizeof(T) > 1) {
        if (!GUARD || i0 + VEC <= n) {
            constexpr int P = piece_of<T, VEC>();
            static_assert(P == 2 || P == 4, "16-byte pieces of 4- or 8-byte elements");
            typedef typename VecOf<T, P>::type V;
#pragma unroll
            for (int q = 0; q < VEC / P; ++q) {
                // (the vector is built by an initializer, not element by element: clang drops the !nontemporal mark of a store
                // whose operand was assembled through subscript assignments — the `nt` bit silently disappears from the ISA)
                V t;
                if constexpr (P == 4) t = V{v[q * P], v[q * P + 1], v[q * P + 2], v[q * P + 3]};
                else t = V{v[q * P], v[q * P + 1]};
#ifdef GYMNET_PROBE_STORE_SC1     // probe builds only (tools/skeleton_floor.hip, mode "sc1"): 16-byte float rows written through (`sc1`)
                if constexpr (NT && sizeof(T) == 4 && P == 4) {
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p + i0 + q * P), "v"(t) : "memory");
                    continue;
                }
#endif
                if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<V *>(p + i0 + q * P));
                else *reinterpret_cast<V *>(p + i0 + q * P) = t;
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        if (!GUARD || i0 + j < n) { if constexpr (NT) __builtin_nontemporal_store(v[j], p + i0 + j); else p[i0 + j] = v[j]; }
}

// done flags: VEC bytes packed into one 16- / 32-bit store
template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void store_u8(uint8_t *__restrict__ p, int64_t i0, int64_t n, const uint8_t (&v)[VEC]) {
    if constexpr (VEC == 2) {
        if (!GUARD || i0 + 2 <= n) {
            const uint16_t w = (uint16_t)((uint16_t)v[0] | ((uint16_t)v[1] << 8));
            if constexpr (NT) __builtin_nontemporal_store(w, reinterpret_cast<uint16_t *>(p + i0));
            else *reinterpret_cast<uint16_t *>(p + i0) = w;
            return;
        }
    }
    if constexpr (VEC == 4) {
        if (!GUARD || i0 + 4 <= n) {
            const uint32_t w = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
            if constexpr (NT) __builtin_nontemporal_store(w, reinterpret_cast<uint32_t *>(p + i0));
            else *reinterpret_cast<uint32_t *>(p + i0) = w;
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        if (!GUARD || i0 + j < n) { if constexpr (NT) __builtin_nontemporal_store(v[j], p + i0 + j); else p[i0 + j] = v[j]; }
}

// the float32 / int32 spellings the kernels use for the streams whose element type never changes (reward, action, counters)
template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void load_f32(const float *__restrict__ p, int64_t i0, int64_t n, float (&v)[VEC]) { load_row<float, VEC, NT, GUARD>(p, i0, n, v); }
template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void store_f32(float *__restrict__ p, int64_t i0, int64_t n, const float (&v)[VEC]) { store_row<float, VEC, NT, GUARD>(p, i0, n, v); }
template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void load_i32(const int32_t *__restrict__ p, int64_t i0, int64_t n, int32_t (&v)[VEC]) { load_row<int32_t, VEC, NT, GUARD>(p, i0, n, v); }
template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void store_i32(int32_t *__restrict__ p, int64_t i0, int64_t n, const int32_t (&v)[VEC]) { store_row<int32_t, VEC, NT, GUARD>(p, i0, n, v); }

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

}  // namespace gymnet
