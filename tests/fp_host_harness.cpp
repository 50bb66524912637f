// Host-side harness: compiles the device field arithmetic (plonky_amd/csrc/fp.cuh, plain C++ when not
// under hipcc) with g++ so the CPU test suite can sweep it against Python integers.
#include <cstddef>
#include <vector>
#include "../plonky_amd/csrc/fp.cuh"
#include "../plonky_amd/csrc/fz.cuh"
#include "../plonky_amd/csrc/ecz.cuh"
#include "../plonky_amd/csrc/glv.cuh"
using namespace plk;

template <class P> static void run(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        Fe<P> x, y, r;
        for (int k = 0; k < P::NL; ++k) { x.v[k] = a[i * P::NL + k]; y.v[k] = b[i * P::NL + k]; }
        switch (op) {
            case 0: r = fe_add<P>(x, y); break;
            case 1: r = fe_sub<P>(x, y); break;
            case 2: r = fe_mul<P>(x, y); break;
            case 3: r = fe_mul_cios<P>(x, y); break;
            case 4: r = fe_sqr<P>(x); break;
            case 5: r = fe_inv<P>(x); break;
            case 6: r = fe_half<P>(x); break;
            case 8: r = fe_inv_eea<P>(x); break;
            case 18: r = fe_inv_safegcd<P>(x); break;
            case 28: r = fe_inv_safegcd_var<P>(x); break;
            // ---- lazy 29-bit-limb arithmetic (fz.cuh): results brought back to canonical words ----
            case 10: r = fz_to_fe_canonical<P>(fz_from_fe<P>(x)); break;
            case 11: r = fz_to_fe_canonical<P>(fz_mul<P>(fz_from_fe<P>(x), fz_from_fe<P>(y))); break;
            case 12: r = fz_to_fe_canonical<P>(fz_sqr<P>(fz_from_fe<P>(x))); break;
            case 13: {  // (x + y - 2y) * 1  through add / dbl / sub / mul by one'
                Fz<P> a = fz_from_fe<P>(x), b = fz_from_fe<P>(y);
                Fz<P> t = fz_sub<P, 2>(fz_add<P>(a, b), fz_dbl<P>(b));
                r = fz_to_fe_canonical<P>(fz_mul<P>(t, fz_one_rprime<P>()));
            } break;
            case 14: r = fz_to_fe_canonical<P>(fz_mul<P>(fz_from_fe<P>(x), fz_const_r_to_rprime<P>())); break;
            case 15: r = fz_to_fe_canonical<P>(fz_mul<P>(fz_from_fe<P>(x), fz_const_rprime_to_r<P>())); break;
            case 16: {  // a longer lazy chain: ((x - y)^2 - (x + y) * y) * (x - 2y + 4p-ish) ...
                Fz<P> a = fz_from_fe<P>(x), b = fz_from_fe<P>(y);
                Fz<P> d = fz_sub<P, 1>(a, b);                    // x - y          (< 3p)
                Fz<P> s = fz_add<P>(a, b);                       // x + y          (< 2p)
                Fz<P> u = fz_sqr<P>(d);                          // (x-y)^2 / R'   (< 2p)
                Fz<P> v = fz_mul<P>(s, b);                       // (x+y) y / R'   (< 2p)
                Fz<P> w = fz_sub<P, 2>(u, fz_dbl<P>(v));         // u - 2v         (< 6p)
                Fz<P> z = fz_mul<P>(w, d);                       // (u - 2v)(x-y) / R'
                r = fz_to_fe_canonical<P>(z);
            } break;
            case 17: {
                Fz<P> m = fz_mul<P>(fz_from_fe<P>(x), fz_from_fe<P>(y));
                r = fe_zero<P>();
                r.v[0] = fz_is_zero_mod_p<P>(m) ? 1u : 0u;
            } break;
            case 19: {  // x + 13 y through lazy additions (< 14p, like the output of a tile's last stage), reduced without a multiplication
                Fz<P> v = fz_from_fe<P>(x);
                const Fz<P> yz = fz_from_fe<P>(y);
                for (int k = 0; k < 13; ++k) v = fz_add<P>(v, yz);
                r = fz_to_fe_canonical<P>(fz_reduce_small<P>(v));
            } break;
            case 20: case 21: case 22: case 23: {
                // two radix-4 steps of the NTT tile on the carry-free forms (ntt.hip tile_stages): the first-step shape on
                // (x, y, y, x), then the general shape on its outputs placed largest-first; twiddles are the normalised x / y
                if constexpr (FzCfg<P>::NZ > 10) { r = fe_zero<P>(); break; } else {
                const Fz<P> a = fz_from_fe<P>(x), b = fz_from_fe<P>(y);
                Fz<P> o[4];
                {
                    const Fz<P> y0 = fz_add_nc<P>(a, b), y1 = fz_sub_nc<P, 1, 29>(a, b), y2 = fz_add_nc<P>(b, a);
                    const Fz<P> y3 = fz_mul<P>(fz_sub_nc<P, 1, 29>(b, a), b);
                    o[1] = fz_add_nc<P>(y1, y3);
                    o[3] = fz_sub_nc<P, 1, 29>(y1, y3);
                    o[0] = fz_add_nc<P>(y0, y2);
                    o[2] = fz_sub_nc<P, 2, 30>(y0, y2);
                }
                Fz<P> x0 = o[3], x1 = o[2], x2 = o[1], x3 = o[0], O[4];
                fz_carry<P>(x0);
                fz_carry<P>(x2);
                x1 = fz_mul<P>(x1, b);
                x3 = fz_mul<P>(x3, b);
                const Fz<P> y0 = fz_add_nc<P>(x0, x1), y1 = fz_sub_nc<P, 1, 29>(x0, x1);
                const Fz<P> y2 = fz_mul<P>(fz_add_nc<P>(x2, x3), b), y3 = fz_mul<P>(fz_sub_nc<P, 1, 29>(x2, x3), a);
                O[1] = fz_add_nc<P>(y1, y3);
                O[3] = fz_sub_nc<P, 1, 29>(y1, y3);
                O[0] = fz_add_nc<P>(y0, y2);
                O[2] = fz_sub_nc<P, 1, 29>(y0, y2);
                bool ok = true;
                for (int k = 0; k < 4; ++k)
                    for (int i = 0; i < FzCfg<P>::NZ; ++i) ok = ok && O[k].l[i] <= FzNcBound<P>::MUL_LIMB_MAX && o[k].l[i] <= FzNcBound<P>::MUL_LIMB_MAX;
                r = ok ? fz_to_fe_canonical<P>(fz_mul<P>(O[op - 20], fz_one_rprime<P>())) : fe_zero<P>();
                }
            } break;
            case 24: {
                // fz_mul_add2 at the edge of its column bound (fz.cuh): every limb below the top one at the largest value the callers may
                // pass (9 limbs: a carried, b with its carries - 2^31 -, c a carry-free negation - 2^30 -, d exactly normalised; 14 limbs:
                // all carried), top limbs as large as a value below 16p leaves them; low bits varied by the inputs
                constexpr int NZ = FzCfg<P>::NZ;
                constexpr bool SMALL = NZ <= 9;
                const uint32_t la = SMALL ? (1u << 29) + 7u : (1u << 29) + (1u << 27), lb = SMALL ? 0x80000000u : la, lc = SMALL ? (1u << 30) : la,
                               ld = (1u << 29) - 1u, top = SMALL ? (1u << 25) : 3u;
                Fz<P> a, b, c, d;
                for (int k = 0; k < NZ - 1; ++k) {
                    a.l[k] = la - (x.v[0] & 7u);
                    b.l[k] = lb - (y.v[0] & 0xffu);
                    c.l[k] = lc - (x.v[2] & 0xffu);
                    d.l[k] = ld - (y.v[1] & 0xffu);
                }
                a.l[NZ - 1] = top + (x.v[1] & 0xffffu);
                b.l[NZ - 1] = c.l[NZ - 1] = top;
                d.l[NZ - 1] = SMALL ? (1u << 22) : 1u;
                r = fz_to_fe_canonical<P>(fz_mul<P>(fz_mul_add2<P>(a, b, c, d), fz_one_rprime<P>()));
            } break;
            case 25: {  // x y - y x' (x' = x + y): fz_mul_sub2, the generic form every XYZZ formula uses
                const Fz<P> xz = fz_from_fe<P>(x), yz = fz_from_fe<P>(y);
                r = fz_to_fe_canonical<P>(fz_mul<P>(fz_mul_sub2<P, 1>(xz, yz, yz, fz_add<P>(xz, yz)), fz_one_rprime<P>()));
            } break;
            default: r = fe_neg<P>(x); break;
        }
        for (int k = 0; k < P::NL; ++k) out[i * P::NL + k] = r.v[k];
    }
}
extern "C" int fp_host_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    switch (field) {
        case 0: run<TweedledeeBaseParams>(op, a, b, out, n); return 0;
        case 1: run<TweedledumBaseParams>(op, a, b, out, n); return 0;
        case 2: run<Bls12377ScalarParams>(op, a, b, out, n); return 0;
        case 3: run<Bls12377BaseParams>(op, a, b, out, n); return 0;
        case 4: run<PallasBaseParams>(op, a, b, out, n); return 0;
        case 5: run<VestaBaseParams>(op, a, b, out, n); return 0;
    }
    return -1;
}

// Lazy XYZZ accumulation (ecz.cuh): acc = sum of +-(x_i, y_i); points given as canonical words of the
// R'-form coordinates; out = X, Y, ZZ, ZZZ (canonical words, R'-form) followed by one word: inf flag.
template <class P> static void ecz_sum(size_t n, const uint32_t* xs, const uint32_t* ys, const uint8_t* negs, uint32_t* out) {
    XyzzZ<P> acc;
    acc.inf = true;
    acc.x = acc.y = acc.zz = acc.zzz = fz_zero<P>();
    out[4 * P::NL] = 0;
    for (size_t i = 0; i < n; ++i) {
        Fe<P> x, y;
        for (int k = 0; k < P::NL; ++k) { x.v[k] = xs[i * P::NL + k]; y.v[k] = ys[i * P::NL + k]; }
        // the accumulation kernel's own step (k_msm_accumulate): the entry in interface words + its sign, Y left uncarried
        // between additions; the carries are moved where a piece would be stored (here: every seventh entry) and at the end
        xyzzz_madd_entry<P>(acc, x, y, negs[i] != 0);
        for (int k = 0; k < FzCfg<P>::NZ; ++k)
            if (acc.y.l[k] > 3u * (1u << 29)) out[4 * P::NL] |= 2u;  // the limb bound of a lazy Y (ecz.cuh)
        if (i % 7 == 6) xyzzz_settle<P>(acc);
    }
    xyzzz_settle<P>(acc);
    for (int k = 0; k < FzCfg<P>::NZ; ++k)
        if (acc.y.l[k] >= (1u << 29) + 8u) out[4 * P::NL] |= 4u;
    Fz<P> one = fz_one_rprime<P>();
    Fe<P> c[4];
    if (acc.inf) { for (auto& e : c) e = fe_zero<P>(); }
    else {
        // multiply by one' to bring every coordinate below 2p before the canonical conversion
        c[0] = fz_to_fe_canonical<P>(fz_mul<P>(acc.x, one));
        c[1] = fz_to_fe_canonical<P>(fz_mul<P>(acc.y, one));
        c[2] = fz_to_fe_canonical<P>(fz_mul<P>(acc.zz, one));
        c[3] = fz_to_fe_canonical<P>(fz_mul<P>(acc.zzz, one));
    }
    for (int j = 0; j < 4; ++j)
        for (int k = 0; k < P::NL; ++k) out[j * P::NL + k] = c[j].v[k];
    out[4 * P::NL] |= acc.inf ? 1u : 0u;  // bit 0: identity; bits 1, 2: a limb bound was exceeded (the test expects 0 / 1 only)
}
// Same sum, but as a balanced tree of full XYZZ additions (xyzzz_add / xyzzz_dbl): the reduction kernels' shape.
template <class P> static void ecz_tree(size_t n, const uint32_t* xs, const uint32_t* ys, const uint8_t* negs, uint32_t* out) {
    std::vector<XyzzZ<P>> v;
    for (size_t i = 0; i < n; ++i) {
        Fe<P> x, y;
        for (int k = 0; k < P::NL; ++k) { x.v[k] = xs[i * P::NL + k]; y.v[k] = ys[i * P::NL + k]; }
        XyzzZ<P> a = xyzzz_identity<P>();
        Fz<P> yz = fz_from_fe<P>(y);
        if (negs[i]) yz = fz_neg_canonical<P>(yz);
        xyzzz_madd<P>(a, fz_from_fe<P>(x), yz);
        v.push_back(a);
    }
    if (v.empty()) v.push_back(xyzzz_identity<P>());
    while (v.size() > 1) {
        std::vector<XyzzZ<P>> w;
        for (size_t i = 0; i + 1 < v.size(); i += 2) w.push_back(xyzzz_add<P>(v[i], v[i + 1]));
        if (v.size() & 1) w.push_back(v.back());
        v.swap(w);
    }
    XyzzZ<P> acc = v[0];
    Fz<P> one = fz_one_rprime<P>();
    Fe<P> c[4];
    if (acc.inf) { for (auto& e : c) e = fe_zero<P>(); }
    else {
        c[0] = fz_to_fe_canonical<P>(fz_mul<P>(acc.x, one));
        c[1] = fz_to_fe_canonical<P>(fz_mul<P>(acc.y, one));
        c[2] = fz_to_fe_canonical<P>(fz_mul<P>(acc.zz, one));
        c[3] = fz_to_fe_canonical<P>(fz_mul<P>(acc.zzz, one));
    }
    for (int j = 0; j < 4; ++j)
        for (int k = 0; k < P::NL; ++k) out[j * P::NL + k] = c[j].v[k];
    out[4 * P::NL] = acc.inf ? 1u : 0u;
}
extern "C" int ecz_host_tree(int field, size_t n, const uint32_t* xs, const uint32_t* ys, const uint8_t* negs, uint32_t* out) {
    switch (field) {
        case 0: ecz_tree<TweedledeeBaseParams>(n, xs, ys, negs, out); return 0;
        case 1: ecz_tree<TweedledumBaseParams>(n, xs, ys, negs, out); return 0;
        case 3: ecz_tree<Bls12377BaseParams>(n, xs, ys, negs, out); return 0;
        case 4: ecz_tree<PallasBaseParams>(n, xs, ys, negs, out); return 0;
        case 5: ecz_tree<VestaBaseParams>(n, xs, ys, negs, out); return 0;
    }
    return -1;
}
extern "C" int ecz_host_sum(int field, size_t n, const uint32_t* xs, const uint32_t* ys, const uint8_t* negs, uint32_t* out) {
    switch (field) {
        case 0: ecz_sum<TweedledeeBaseParams>(n, xs, ys, negs, out); return 0;
        case 1: ecz_sum<TweedledumBaseParams>(n, xs, ys, negs, out); return 0;
        case 3: ecz_sum<Bls12377BaseParams>(n, xs, ys, negs, out); return 0;
        case 4: ecz_sum<PallasBaseParams>(n, xs, ys, negs, out); return 0;
        case 5: ecz_sum<VestaBaseParams>(n, xs, ys, negs, out); return 0;
    }
    return -1;
}

// GLV split (glv.cuh): canonical scalars in, (magnitude | sign in bit 255) pairs out
template <class G> static void glv_run(const uint32_t* k, uint32_t* k1, uint32_t* k2, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t a[8], b[8], c[8];
        for (int t = 0; t < 8; ++t) a[t] = k[8 * i + t];
        glv_split<G>(a, b, c);
        for (int t = 0; t < 8; ++t) { k1[8 * i + t] = b[t]; k2[8 * i + t] = c[t]; }
    }
}
extern "C" int glv_host_split(int curve, const uint32_t* k, uint32_t* k1, uint32_t* k2, size_t n) {
    switch (curve) {
        case 0: glv_run<TweedledeeGlv>(k, k1, k2, n); return 0;
        case 1: glv_run<TweedledumGlv>(k, k1, k2, n); return 0;
        case 3: glv_run<PallasGlv>(k, k1, k2, n); return 0;
        case 4: glv_run<VestaGlv>(k, k1, k2, n); return 0;
    }
    return -1;
}
