// Included by a kernel source AFTER the defaults of its development switches (knock-outs, traces, A/B experiments).
// A build with any switch set is a DEV build: it must say so (-DDMVS_DEV_BUILD, or it does not compile), and it then exports
// dmvs_dev_build(), which dmvsnet_amd/_lib.py refuses to load unless DMVS_ALLOW_DEV_BUILD=1 -- a stray -D flag can no longer
// produce a silently wrong library with the product's ABI version.
#if (defined(DMVS_KO) && DMVS_KO) || (defined(DMVS_X) && DMVS_X) || (defined(DMVS_WKO) && DMVS_WKO) ||                         \
    (defined(DMVS_WINO_TAU) && DMVS_WINO_TAU) || (defined(DMVS_C8_KO) && DMVS_C8_KO) || defined(C8_FAKE_ALIGNED) ||           \
    (defined(C8_NS) && C8_NS != 4) || (defined(DMVS_CONV1_CI) && DMVS_CONV1_CI != 1) || (defined(DMVS_K3R_LW) && DMVS_K3R_LW != 8) || (defined(DMVS_K3R_RING) && DMVS_K3R_RING != 3) || (defined(DMVS_K3Z_RING) && DMVS_K3Z_RING != 2) || (defined(DMVS_ZKO) && DMVS_ZKO) || (defined(DMVS_TILE_AUX) && DMVS_TILE_AUX != 0) || defined(DMVS_K3_TRACE) ||             \
    defined(DMVS_K3R_TRACE) || defined(DMVS_Q4_TRACE)
#ifndef DMVS_DEV_BUILD
#error "development switches (DMVS_KO / DMVS_X / DMVS_WKO / DMVS_C8_KO / C8_* / *_TRACE ...) need -DDMVS_DEV_BUILD: the library is then marked as a dev build"
#endif
#endif
#ifdef DMVS_DEV_BUILD
#ifndef DMVS_DEV_BUILD_SYMBOL
#define DMVS_DEV_BUILD_SYMBOL
extern "C" __attribute__((weak, visibility("default"))) int dmvs_dev_build(void) { return 1; }
#endif
#endif
