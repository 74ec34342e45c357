// TEST INFRASTRUCTURE ONLY (see ../hip/hip_runtime.h): host stand-ins for the rocPRIM device primitives the overlap library
// calls, with rocPRIM's calling convention (a first call with temporary_storage == nullptr returns the size needed).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {

template <class T> struct plus {
    T operator()(const T &a, const T &b) const { return a + b; }
};
template <class T> struct maximum {
    T operator()(const T &a, const T &b) const { return a < b ? b : a; }
};

template <class It, class F> struct transform_iterator {
    It it;
    F f;
    auto operator[](size_t i) const { return f(it[i]); }
};
template <class It, class F> transform_iterator<It, F> make_transform_iterator(It it, F f) { return transform_iterator<It, F>{it, f}; }

namespace detail {
inline bool size_query(void *tmp, size_t &bytes) {
    if (tmp) return false;
    bytes = 256;
    return true;
}
template <class K> inline unsigned long long key_bits(K k, unsigned b0, unsigned b1) {
    const unsigned long long v = (unsigned long long)k >> b0;
    const unsigned w = b1 - b0;
    return w >= 64 ? v : v & ((1ull << w) - 1ull);
}
}  // namespace detail

template <class K, class V>
hipError_t radix_sort_pairs(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, size_t n, unsigned b0 = 0,
                            unsigned b1 = 8 * sizeof(K), hipStream_t = nullptr) {
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), (size_t)0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return detail::key_bits(kin[a], b0, b1) < detail::key_bits(kin[b], b0, b1); });
    std::vector<K> k(n);
    std::vector<V> v(n);
    for (size_t i = 0; i < n; i++) k[i] = kin[idx[i]], v[i] = vin[idx[i]];
    std::copy(k.begin(), k.end(), kout);
    std::copy(v.begin(), v.end(), vout);
    return hipSuccess;
}

template <class K>
hipError_t radix_sort_keys(void *tmp, size_t &bytes, const K *kin, K *kout, size_t n, unsigned b0 = 0, unsigned b1 = 8 * sizeof(K),
                           hipStream_t = nullptr) {
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    std::vector<K> k(kin, kin + n);
    std::stable_sort(k.begin(), k.end(), [&](K a, K b) { return detail::key_bits(a, b0, b1) < detail::key_bits(b, b0, b1); });
    std::copy(k.begin(), k.end(), kout);
    return hipSuccess;
}

template <class K, class U, class C, class N>
hipError_t run_length_encode(void *tmp, size_t &bytes, const K *in, unsigned int n, U *uniq, C *cnt, N *n_runs, hipStream_t = nullptr) {
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    size_t r = 0;
    for (unsigned int i = 0; i < n;) {
        unsigned int j = i;
        while (j < n && in[j] == in[i]) j++;
        uniq[r] = in[i];
        cnt[r] = (C)(j - i);
        r++;
        i = j;
    }
    *n_runs = (N)r;
    return hipSuccess;
}

template <class In, class Out, class T, class Op>
hipError_t exclusive_scan(void *tmp, size_t &bytes, In in, Out out, T init, size_t n, Op op, hipStream_t = nullptr) {
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    T acc = init;
    for (size_t i = 0; i < n; i++) {
        const T v = (T)in[i];
        out[i] = acc;
        acc = op(acc, v);
    }
    return hipSuccess;
}

template <class In, class Out, class Op>
hipError_t inclusive_scan(void *tmp, size_t &bytes, In in, Out out, size_t n, Op op, hipStream_t = nullptr) {
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    if (!n) return hipSuccess;
    auto acc = in[0];
    out[0] = acc;
    for (size_t i = 1; i < n; i++) {
        acc = op(acc, in[i]);
        out[i] = acc;
    }
    return hipSuccess;
}

}  // namespace rocprim
