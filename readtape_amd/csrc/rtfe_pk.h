// rtfe_pk.h — two int16 samples per 32-bit register ("packed" lanes: lo = one head, hi = its neighbour).
// On gfx950 every helper is one VOP3P instruction (v_pk_max_i16, v_pk_sub_i16 clamp, ...).  The second
// definition exists only so that tests/cpu_emul can compile the unmodified kernel sources with g++.
#pragma once
#include <stdint.h>

namespace rtfe {

#ifndef RTFE_CPU_EMUL
typedef short          pk_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u2 __attribute__((ext_vector_type(2)));
#define RTFE_PK __device__ __forceinline__
RTFE_PK uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b))); }
RTFE_PK uint32_t pk_min(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b))); }
RTFE_PK uint32_t pk_subs(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b))); }   // saturating a - b
RTFE_PK uint32_t pk_adds(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b))); }
RTFE_PK uint32_t pk_addu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (pk_u2)(__builtin_bit_cast(pk_u2, a) + __builtin_bit_cast(pk_u2, b))); }                     // wrapping
RTFE_PK uint32_t pk_maxu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_u2, a), __builtin_bit_cast(pk_u2, b))); }
#else
#define RTFE_PK static inline
static inline int pk__lo(uint32_t a) { return (int16_t)(a & 0xffff); }
static inline int pk__hi(uint32_t a) { return (int16_t)(a >> 16); }
static inline uint32_t pk__mk(int lo, int hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); }
static inline int pk__sat(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
RTFE_PK uint32_t pk_max(uint32_t a, uint32_t b) { return pk__mk(pk__lo(a) > pk__lo(b) ? pk__lo(a) : pk__lo(b), pk__hi(a) > pk__hi(b) ? pk__hi(a) : pk__hi(b)); }
RTFE_PK uint32_t pk_min(uint32_t a, uint32_t b) { return pk__mk(pk__lo(a) < pk__lo(b) ? pk__lo(a) : pk__lo(b), pk__hi(a) < pk__hi(b) ? pk__hi(a) : pk__hi(b)); }
RTFE_PK uint32_t pk_subs(uint32_t a, uint32_t b) { return pk__mk(pk__sat(pk__lo(a) - pk__lo(b)), pk__sat(pk__hi(a) - pk__hi(b))); }
RTFE_PK uint32_t pk_adds(uint32_t a, uint32_t b) { return pk__mk(pk__sat(pk__lo(a) + pk__lo(b)), pk__sat(pk__hi(a) + pk__hi(b))); }
RTFE_PK uint32_t pk_addu(uint32_t a, uint32_t b) { return ((a & 0xffff) + (b & 0xffff) & 0xffff) | (((a >> 16) + (b >> 16)) << 16); }
RTFE_PK uint32_t pk_maxu(uint32_t a, uint32_t b) { const uint32_t al = a & 0xffff, bl = b & 0xffff, ah = a >> 16, bh = b >> 16; return (al > bl ? al : bl) | ((ah > bh ? ah : bh) << 16); }
#endif

RTFE_PK uint32_t pk_dup(int v) { return (uint32_t)(uint16_t)v | ((uint32_t)(uint16_t)v << 16); }
constexpr uint32_t kPkSigns = 0x80008000u;      // the sign bit of each half: "is the difference negative" masks

}  // namespace rtfe
