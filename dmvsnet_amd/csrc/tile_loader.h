// LDS-direct tile staging shared by the direct (K2) and MFMA (K3) convolution kernels.
#pragma once
#include "common.h"

// Stage CI_CH channels of an input tile (+halo, zero padded) in LDS with LDS-direct buffer loads
// (buffer_load_dword ... offen lds): no VGPR round trip, no ds_write, fully asynchronous.
//  * one tile row (fixed c, z, y) per wave-instruction: lane l fetches x = ix0 + l and the hardware writes it to
//    LDS at (wave-uniform row base) + 4*l, i.e. exactly the [ci][z][y][x] tile layout; 256-byte coalesced reads;
//  * zero padding comes from the buffer descriptor's range check: an element outside the volume gets a byte
//    offset >= 2^31 > num_records and the load returns 0 -- every load is unconditional straight-line code
//    (a predicated load becomes an exec-masked block with a vmcnt(0) behind it and serialises on HBM latency:
//    measured 3.6x slower layers);
//  * fully unrolled over (c, z, y-slot); the row base is scalar arithmetic, per lane one add + one or;
//  * wave w owns rows y = w, w+4, ...; a slot past the last row re-loads the last row (harmless duplicate);
//  * rows wider than 64 floats (stride-2 tiles: 65) get their tail columns through VGPRs, lanes = rows.
// `rsrc` describes the CI_CH channels of THIS chunk only (base = first channel of the chunk), so offsets stay
// below both invalid markers as long as CI_CH*D*H*W < 2^28 elements (checked by the launchers); ci0 is unused.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifndef DMVS_TILE_AUX
#define DMVS_TILE_AUX 0   /* cache policy of the 16-byte tile loads.  r05 same-box A/B of the whole forward (4 alternating runs): default 89.16, nt (2) 83.22 -- the halo rows neighbouring tiles re-read no longer stay in the XCD's L2 --, sc0 (1) 89.17 depth-maps/s */
#endif

template <int CI_CH, int IZ, int IY, int IX, int IXP, int PS, bool NEG>
// cs / zs: element strides of a channel / a depth slice; 0 = the planar [C][D][H][W] defaults (D * H * W, H * W).  The conv
// kernels pass other strides for DMVS_IN_VIEWS (FeatureNet's first layer reading the loader's [V][3][H][W] images directly).
__device__ __forceinline__ void load_tile(int aD, int aH, int aW, __amdgpu_buffer_rsrc_t rsrc, float* tile, int ci0,
                                          int iz0, int iy0, int ix0, int wave, int lane, int cs = 0, int zs = 0) {
    constexpr int MW = IX < 64 ? IX : 64;
    constexpr int YI = (IY + 3) / 4;
    // row-invalid and x-invalid markers are different bits so that their SUM cannot wrap back into range
    constexpr unsigned kInvalid = 0x80000000u, kInvalidX = 0x40000000u;
    const int plane = zs ? zs : aH * aW, vol = cs ? cs : aD * aH * aW;
    const int gx = ix0 + lane;
    const bool xin = (!NEG || gx >= 0) && gx < aW;
    const unsigned gx4 = xin ? (unsigned)gx * 4u : kInvalidX;
    int yoff[YI], ly[YI];
    bool yin[YI];
#pragma unroll
    for (int k = 0; k < YI; ++k) {
        const int y = min(wave + 4 * k, IY - 1), gy = iy0 + y;
        yin[k] = (!NEG || gy >= 0) && gy < aH;
        yoff[k] = gy * aW;
        ly[k] = y * IXP;
    }
    if (lane < MW) {  // ONE exec region: lanes past the row end must not spill into the next LDS row
#pragma unroll
        for (int c = 0; c < CI_CH; ++c) {
#pragma unroll
            for (int z = 0; z < IZ; ++z) {
                const int gz = iz0 + z;
                const bool zin = (!NEG || gz >= 0) && gz < aD;
                const int cz = c * vol + gz * plane;  // rsrc is based at the chunk's first channel
#pragma unroll
                for (int k = 0; k < YI; ++k) {
#if defined(DMVS_KO) && (DMVS_KO & 16)
                    const unsigned rb = kInvalid;
#else
                    const unsigned rb = (zin && yin[k]) ? (unsigned)(cz + yoff[k]) * 4u : kInvalid;  // scalar
#endif
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(tile + c * PS + z * IY * IXP + ly[k]), 4,
                                                             rb + gx4, 0, 0, 0);
                }
            }
        }
    }
    if (IX > 64) {
        constexpr int NROWS = CI_CH * IZ * IY;
        constexpr int NT = (IX - 64) * NROWS;
#pragma unroll 2
        for (int idx = wave * 64 + lane; idx < NT; idx += 256) {
            const int r = idx % NROWS, x = 64 + idx / NROWS;
            const int y = r % IY, z = (r / IY) % IZ, c = r / (IY * IZ);
            const int gz = iz0 + z, gy = iy0 + y, gxx = ix0 + x;
            const bool ok = gz >= 0 && gz < aD && gy >= 0 && gy < aH && gxx >= 0 && gxx < aW;
            const unsigned off = ok ? (unsigned)(c * vol + gz * plane + gy * aW + gxx) * 4u : kInvalid;
            tile[c * PS + (z * IY + y) * IXP + x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
        }
    }
}


// 16-byte variant (gfx950: buffer_load_dwordx4 ... lds): a wave-instruction moves RPI = 64 / LPR whole tile rows
// of LPR 16-byte pieces -- 4-6x fewer load instructions than one row of dwords each.  An LDS-direct load costs the
// issuing wave ~60-100 cycles whatever its width (measured: loads issued with out-of-range offsets, no traffic,
// still cost 70 % of conv1's load phase), so instruction count, not bytes, is what the loader has to minimise.
//  * the tile row starts at a 16-byte aligned x (ix0a, a multiple of 4) and is IXP = 4 * LPR floats wide, dense:
//    LDS layout [c][r = z * IY + y][IXP], channel stride PS (free: padded by the caller for bank spreading);
//  * requires W % 4 == 0 and a 16-byte aligned base, so that a piece is entirely inside or outside its row;
//  * lanes past the last row of an instruction are switched off (exec), not sent out of range: they would zero the
//    first pieces of the next channel's plane.
template <int CI_CH, int IZ, int IY, int LPR, int PS>
__device__ __forceinline__ void load_tile4(int aD, int aH, int aW, __amdgpu_buffer_rsrc_t rsrc, float* tile, int iz0,
                                           int iy0, int ix0a, int wave, int lane, int cs = 0, int zs = 0) {
    constexpr int RPI = 64 / LPR;             // rows per wave-instruction
    constexpr int NR = IZ * IY;               // rows per channel
    constexpr int G = (NR + RPI - 1) / RPI;   // instructions per channel
    constexpr int IXP = 4 * LPR;
    constexpr unsigned kInvalid = 0x80000000u;
    const int plane = zs ? zs : aH * aW, vol = cs ? cs : aD * aH * aW;
    const int lr = lane / LPR, x4 = lane - lr * LPR;
    const int gx = ix0a + 4 * x4;
    const bool xin = (unsigned)gx < (unsigned)aW;
#pragma unroll
    for (int k = 0; k < (G + 3) / 4; ++k) {
        const int g = min(wave + 4 * k, G - 1);  // scalar; a slot past the end repeats the last instruction
        const int r = g * RPI + lr;
        const int z = r / IY, y = r - z * IY;
        const int gz = iz0 + z, gy = iy0 + y;
        const bool ok = xin && (unsigned)gz < (unsigned)aD && (unsigned)gy < (unsigned)aH;
        const int rel = gz * plane + gy * aW + gx;
        if (lr < RPI && r < NR) {
#pragma unroll
            for (int c = 0; c < CI_CH; ++c) {
                const unsigned off = ok ? (unsigned)(c * vol + rel) * 4u : kInvalid;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(tile + c * PS + g * RPI * IXP), 16, off, 0, 0, DMVS_TILE_AUX);
            }
        }
    }
}
