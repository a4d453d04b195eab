/* Prints sizeof / offsetof of every struct of include/s2k.h as "name value" lines (tests/test_abi.py compares them with the
 * ctypes mirrors in rust-seq2kminmers_amd/__init__.py).  Plain C11: the header must be usable from C. */
#include "s2k.h"
#include <stddef.h>
#include <stdio.h>
#define SZ(T) printf("sizeof." #T " %zu\n", sizeof(T))
#define OFF(T, f) printf(#T "." #f " %zu %zu\n", offsetof(T, f), sizeof(((T *)0)->f))
int main(void) {
    SZ(s2k_params); OFF(s2k_params, l); OFF(s2k_params, k); OFF(s2k_params, density); OFF(s2k_params, mode); OFF(s2k_params, flags);
    SZ(s2k_counts); OFF(s2k_counts, n_reads); OFF(s2k_counts, n_bases); OFF(s2k_counts, n_minimizers); OFF(s2k_counts, n_kminmers);
    OFF(s2k_counts, xor_hash); OFF(s2k_counts, hash_bound); OFF(s2k_counts, path);
    SZ(s2k_result); OFF(s2k_result, n_reads); OFF(s2k_result, n_kminmers); OFF(s2k_result, km_off); OFF(s2k_result, hash);
    OFF(s2k_result, start); OFF(s2k_result, end); OFF(s2k_result, rev); OFF(s2k_result, n_minimizers); OFF(s2k_result, mn_off);
    OFF(s2k_result, mn_j); OFF(s2k_result, mn_jend); OFF(s2k_result, mn_hash); OFF(s2k_result, counts); OFF(s2k_result, _owner);
    SZ(s2k_device_out); OFF(s2k_device_out, km_capacity); OFF(s2k_device_out, km_off); OFF(s2k_device_out, hash);
    OFF(s2k_device_out, start); OFF(s2k_device_out, end); OFF(s2k_device_out, rev); OFF(s2k_device_out, mn_capacity);
    OFF(s2k_device_out, mn_off); OFF(s2k_device_out, mn_j); OFF(s2k_device_out, mn_jend); OFF(s2k_device_out, mn_hash);
    printf("enum.S2K_MODE_REGULAR %d\nenum.S2K_MODE_HPC %d\nenum.S2K_MODE_SIMD %d\nenum.S2K_MODE_HPCSIMD %d\n", S2K_MODE_REGULAR, S2K_MODE_HPC,
           S2K_MODE_SIMD, S2K_MODE_HPCSIMD);
    printf("enum.S2K_FLAG_WANT_MINIMIZERS %u\nenum.S2K_FLAG_FORCE_SERIAL %u\nenum.S2K_FLAG_NO_PACK2 %u\nenum.S2K_HPC_RLE_ALPHABET %u\nenum.S2K_ABI_VERSION %d\n",
           (unsigned)S2K_FLAG_WANT_MINIMIZERS, (unsigned)S2K_FLAG_FORCE_SERIAL, (unsigned)S2K_FLAG_NO_PACK2, (unsigned)S2K_HPC_RLE_ALPHABET, S2K_ABI_VERSION);
    printf("enum.S2K_ERR_NO_DEVICE %d\nenum.S2K_ERR_CAPACITY %d\nenum.S2K_FLAG_LEGACY_PATH %u\n", S2K_ERR_NO_DEVICE, S2K_ERR_CAPACITY, (unsigned)S2K_FLAG_LEGACY_PATH);
    return 0;
}
