// oracle/plk_oracle.cpp -- CPU restatement of the Plonky NTT + MSM hot path.
//
// TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
// bench.py may load liboracle.so.  Nothing under plonky_amd/ links, imports or calls it; the
// product path fails loudly if the HIP library is missing instead of falling back to this.
//
// What it is: a C++17 restatement (u64 limbs, unsigned __int128) of the *algorithms* of the
// reference crate 0xPolygonZero/plonky, function by function, each citing the reference
// file:line it follows.  The reference is Rust (nightly) and cannot be built in this image
// (no cargo/rustc), so this restatement + oracle/bigint_ref.py (independent big-int maths)
// are the parity oracle.  Parity pinning: tests/test_oracle_kats.py checks this file against
// every constant, KAT and unit-test vector the reference holds for the path
// and against bigint_ref.py on seeded inputs.
//
// It is also the "port" CPU baseline of bench.py: same algorithm structure as the reference's
// Rayon path (per-layer barrier NTT with 2000-pair chunks; Yao MSM with per-generator power
// tables, serial digit scatter, 80-digit chunks of batch-inversion affine multi-summation,
// serial tail).  Always label it "C++ restatement of the reference algorithm", never "plonky".

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

typedef uint64_t u64;
typedef unsigned __int128 u128;

// =============================================================================================
// bigint primitives -- src/bigint/bigint_arithmetic.rs
// =============================================================================================
template <size_t N> using Limbs = std::array<u64, N>;

// bigint_arithmetic.rs:11-21
template <size_t N> static inline int cmp(const Limbs<N>& a, const Limbs<N>& b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
// bigint_arithmetic.rs:25-38
template <size_t N> static inline Limbs<N> add_no_overflow(const Limbs<N>& a, const Limbs<N>& b) {
    Limbs<N> s;
    u64 carry = 0;
    for (size_t i = 0; i < N; ++i) {
        u128 t = (u128)a[i] + b[i] + carry;
        s[i] = (u64)t;
        carry = (u64)(t >> 64);
    }
    return s;
}
// bigint_arithmetic.rs:42-55
template <size_t N> static inline Limbs<N> sub(const Limbs<N>& a, const Limbs<N>& b) {
    Limbs<N> d;
    u64 borrow = 0;
    for (size_t i = 0; i < N; ++i) {
        u128 t = (u128)a[i] - b[i] - borrow;
        d[i] = (u64)t;
        borrow = (u64)(t >> 64) & 1;
    }
    return d;
}
// bigint_arithmetic.rs:70-79
template <size_t N> static inline Limbs<N> mul2(const Limbs<N>& x) {
    Limbs<N> r;
    r[0] = x[0] << 1;
    for (size_t i = 1; i < N; ++i) r[i] = (x[i] << 1) | (x[i - 1] >> 63);
    return r;
}
// bigint_arithmetic.rs:84-91
template <size_t N> static inline Limbs<N> div2(const Limbs<N>& x) {
    Limbs<N> r;
    for (size_t i = 0; i < N - 1; ++i) r[i] = (x[i] >> 1) | (x[i + 1] << 63);
    r[N - 1] = x[N - 1] >> 1;
    return r;
}
template <size_t N> static inline bool is_zero_l(const Limbs<N>& x) {
    u64 o = 0;
    for (size_t i = 0; i < N; ++i) o |= x[i];
    return o == 0;
}
template <size_t N> static inline bool is_one_l(const Limbs<N>& x) {
    u64 o = x[0] ^ 1;
    for (size_t i = 1; i < N; ++i) o |= x[i];
    return o == 0;
}

// src/bigint/bigint_inverse.rs:6-55 (binary extended Euclid, "Algorithm 16")
template <size_t N> static Limbs<N> nonzero_multiplicative_inverse(const Limbs<N>& a, const Limbs<N>& order) {
    Limbs<N> u = a, v = order, b{}, c{};
    b[0] = 1;
    while (!is_one_l(u) && !is_one_l(v)) {
        while ((u[0] & 1) == 0) {
            u = div2(u);
            if (b[0] & 1) b = add_no_overflow(b, order);
            b = div2(b);
        }
        while ((v[0] & 1) == 0) {
            v = div2(v);
            if (c[0] & 1) c = add_no_overflow(c, order);
            c = div2(c);
        }
        if (cmp(u, v) < 0) {
            v = sub(v, u);
            if (cmp(c, b) < 0) c = add_no_overflow(c, order);
            c = sub(c, b);
        } else {
            u = sub(u, v);
            if (cmp(b, c) < 0) b = add_no_overflow(b, order);
            b = sub(b, c);
        }
    }
    return is_one_l(u) ? b : c;
}

// =============================================================================================
// Field parameter packs.  Every literal is a constant of the reference (file:line given);
// tests re-derive each from the modulus with Python integers.
// =============================================================================================
struct TweedledeeBaseP {  // src/field/tweedledee_base.rs:20-64, :117-171
    static constexpr int N = 4;
    static constexpr int BITS = 255, TWO_ADICITY = 34;
    static constexpr bool MONTY_SQUARE = true;   // :205-209 overrides square() with monty_square
    static constexpr bool SHIFT_DOUBLE = false;  // double/triple are full multiplies (field.rs:180-188)
    static constexpr u64 ORDER[4] = {9524180637049683969ull, 255193519543715529ull, 0ull, 4611686018427387904ull};
    static constexpr u64 R[4] = {8320946236270051325ull, 17681163515078405027ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 R2[4] = {9625875206237061136ull, 9085631154807722544ull, 17636350113745641634ull, 56485833733595155ull};
    static constexpr u64 R3[4] = {11971961131424865118ull, 6311318431551332850ull, 14638507591886519234ull, 739379759776372087ull};
    static constexpr u64 MU = 9524180637049683967ull;
    static constexpr u64 TWO[4] = {7117711835490418681ull, 16660389436903542909ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 THREE[4] = {5914477434710786037ull, 15639615358728680791ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 GENERATOR[4] = {3508008633151520749ull, 13598067202378956555ull, 18446744073709551615ull, 4611686018427387903ull};  // FIVE
    static constexpr u64 T[4] = {9524180637049683969ull, 255193519543715529ull, 0ull, 4611686017353646080ull};
};
struct TweedledumBaseP {  // src/field/tweedledum_base.rs:20-64, :117-171
    static constexpr int N = 4;
    static constexpr int BITS = 255, TWO_ADICITY = 33;
    static constexpr bool MONTY_SQUARE = true;
    static constexpr bool SHIFT_DOUBLE = false;
    static constexpr u64 ORDER[4] = {11619397960441266177ull, 255193519591741881ull, 0ull, 4611686018427387904ull};
    static constexpr u64 R[4] = {2035294266095304701ull, 17681163514934325971ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 R2[4] = {2885853259929485328ull, 10494584067553537908ull, 15959394653775906393ull, 56485833754855950ull};
    static constexpr u64 R3[4] = {11023471670160566071ull, 18013763770685241468ull, 7203328081223416457ull, 2412999303287602290ull};
    static constexpr u64 MU = 11619397960441266175ull;
    static constexpr u64 TWO[4] = {10897934645458894841ull, 16660389436567358444ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 THREE[4] = {1313830951112933365ull, 15639615358200390918ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 GENERATOR[4] = {592367636130562029ull, 13598067201466455865ull, 18446744073709551615ull, 4611686018427387903ull};  // FIVE
    static constexpr u64 T[4] = {11619397960441266177ull, 255193519591741881ull, 0ull, 4611686016279904256ull};
};
struct Bls12377ScalarP {  // src/field/bls12_377_scalar.rs:23-41, :154-173
    static constexpr int N = 4;
    static constexpr int BITS = 253, TWO_ADICITY = 47;
    static constexpr bool MONTY_SQUARE = false;  // no square override: square() = self * self (field.rs:171-173)
    static constexpr bool SHIFT_DOUBLE = false;
    static constexpr u64 ORDER[4] = {725501752471715841ull, 6461107452199829505ull, 6968279316240510977ull, 1345280370688173398ull};
    static constexpr u64 R[4] = {9015221291577245683ull, 8239323489949974514ull, 1646089257421115374ull, 958099254763297437ull};
    static constexpr u64 R2[4] = {2726216793283724667ull, 14712177743343147295ull, 12091039717619697043ull, 81024008013859129ull};
    static constexpr u64 R3[4] = {7656847007262524748ull, 7083357369969088153ull, 12818756329091487507ull, 432872940405820890ull};
    static constexpr u64 MU = 725501752471715839ull;
    static constexpr u64 TWO[4] = {17304940830682775525ull, 10017539527700119523ull, 14770643272311271387ull, 570918138838421475ull};
    static constexpr u64 THREE[4] = {7147916296078753751ull, 11795755565450264533ull, 9448453213491875784ull, 183737022913545514ull};
    static constexpr u64 GENERATOR[4] = {1855201571499933546ull, 8511318076631809892ull, 6222514765367795509ull, 1122129207579058019ull};
    static constexpr u64 T[4] = {725501752471715841ull, 6461107452199829505ull, 6968279316240510977ull, 1345280370688042326ull};
};
struct Bls12377BaseP {  // src/field/bls12_377_base.rs:23-46, :169-205
    static constexpr int N = 6;
    static constexpr int BITS = 377, TWO_ADICITY = 46;
    static constexpr bool MONTY_SQUARE = false;
    static constexpr bool SHIFT_DOUBLE = true;   // :229-253 double()/triple() by shifts
    static constexpr u64 ORDER[6] = {9586122913090633729ull, 1660523435060625408ull, 2230234197602682880ull, 1883307231910630287ull, 14284016967150029115ull, 121098312706494698ull};
    static constexpr u64 R[6] = {202099033278250856ull, 5854854902718660529ull, 11492539364873682930ull, 8885205928937022213ull, 5545221690922665192ull, 39800542322357402ull};
    static constexpr u64 R2[6] = {13224372171368877346ull, 227991066186625457ull, 2496666625421784173ull, 13825906835078366124ull, 9475172226622360569ull, 30958721782860680ull};
    static constexpr u64 R3[6] = {6349885463227391520ull, 16505482940020594053ull, 3163973454937060627ull, 7650090842119774734ull, 4571808961100582073ull, 73846176275226021ull};
    static constexpr u64 MU = 9586122913090633727ull;
    static constexpr u64 TWO[6] = {404198066556501712ull, 11709709805437321058ull, 4538334656037814244ull, 17770411857874044427ull, 11090443381845330384ull, 79601084644714804ull};
    static constexpr u64 THREE[6] = {606297099834752568ull, 17564564708155981587ull, 16030874020911497174ull, 8208873713101515024ull, 16635665072767995577ull, 119401626967072206ull};
    static constexpr u64 GENERATOR[6] = {9871116327010172167ull, 9167007004823125620ull, 18338974479346628539ull, 5649234265355377548ull, 13442091487463296847ull, 77904398905292312ull};  // FIVE
    static constexpr u64 T[6] = {9586122913090633729ull, 1660523435060625408ull, 2230234197602682880ull, 1883307231910630287ull, 14284016967150029115ull, 121098312706232554ull};
};
struct PallasBaseP {  // src/field/pallas_base.rs:20-60, :117-171
    static constexpr int N = 4;
    static constexpr int BITS = 255, TWO_ADICITY = 32;
    static constexpr bool MONTY_SQUARE = true;   // :208-212 overrides square() with monty_square
    static constexpr bool SHIFT_DOUBLE = false;
    static constexpr u64 ORDER[4] = {11037532056220336129ull, 2469829653914515739ull, 0ull, 4611686018427387904ull};
    static constexpr u64 R[4] = {3780891978758094845ull, 11037255111966004397ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 R2[4] = {10122100416058490895ull, 15551789045973377255ull, 8617542898466512152ull, 679271340751763220ull};
    static constexpr u64 R3[4] = {17403498412575166713ull, 17773050464821424593ull, 16108549121152898092ull, 3090323811697793296ull};
    static constexpr u64 MU = 11037532056220336127ull;
    static constexpr u64 TWO[4] = {14970995975005405177ull, 1157936496307941438ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 THREE[4] = {7714355897543163893ull, 9725361954359430096ull, 18446744073709551614ull, 4611686018427387903ull};
    static constexpr u64 GENERATOR[4] = {11647819816328232941ull, 8413468796752855795ull, 18446744073709551613ull, 4611686018427387903ull};  // FIVE
    static constexpr u64 T[4] = {11037532056220336129ull, 2469829653914515739ull, 0ull, 4611686014132420608ull};
};
struct VestaBaseP {  // src/field/vesta_base.rs:20-60, :117-171
    static constexpr int N = 4;
    static constexpr int BITS = 255, TWO_ADICITY = 32;
    static constexpr bool MONTY_SQUARE = true;   // :208-212 overrides square() with monty_square
    static constexpr bool SHIFT_DOUBLE = false;
    static constexpr u64 ORDER[4] = {10108024940646105089ull, 2469829653919213789ull, 0ull, 4611686018427387904ull};
    static constexpr u64 R[4] = {6569413325480787965ull, 11037255111951910247ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 R2[4] = {18200867980676431887ull, 7474641938123724515ull, 9200329640471491984ull, 679271340771891881ull};
    static constexpr u64 R3[4] = {39197710403612236ull, 16229805722976916262ull, 9871554806900181859ull, 566775843421393608ull};
    static constexpr u64 MU = 10108024940646105087ull;
    static constexpr u64 TWO[4] = {3030801710315470841ull, 1157936496275055089ull, 18446744073709551615ull, 4611686018427387903ull};
    static constexpr u64 THREE[4] = {17938934168859705333ull, 9725361954307751546ull, 18446744073709551614ull, 4611686018427387903ull};
    static constexpr u64 GENERATOR[4] = {10861710938529071085ull, 8413468796663592846ull, 18446744073709551613ull, 4611686018427387903ull};  // FIVE
    static constexpr u64 T[4] = {10108024940646105089ull, 2469829653919213789ull, 0ull, 4611686014132420608ull};
};

template <class P, int N = P::N> static inline Limbs<N> L(const u64 (&a)[N]) {
    Limbs<N> r;
    for (size_t i = 0; i < N; ++i) r[i] = a[i];
    return r;
}

// =============================================================================================
// Field element -- src/field/monty.rs (4 limbs), bls12_377_base.rs:58-98 (6 limbs), field.rs
// =============================================================================================
template <class P> struct Fp {
    using Params = P;
    static constexpr int N = P::N;
    Limbs<N> limbs;  // Montgomery form, fully reduced (tweedledee_base.rs:14-18)

    static Limbs<N> order() { return L<P>(P::ORDER); }
    static Fp zero() { Fp r; r.limbs.fill(0); return r; }
    static Fp one() { Fp r; r.limbs = L<P>(P::R); return r; }
    static Fp two() { Fp r; r.limbs = L<P>(P::TWO); return r; }
    static Fp three() { Fp r; r.limbs = L<P>(P::THREE); return r; }

    bool is_zero() const { return is_zero_l(limbs); }
    bool operator==(const Fp& o) const { return limbs == o.limbs; }
    bool operator!=(const Fp& o) const { return !(limbs == o.limbs); }

    // monty.rs:67-107 / bls12_377_base.rs:58-98 -- CIOS with a wrap-around (N+1)-word window
    static Limbs<N> monty_multiply(const Limbs<N>& a, const Limbs<N>& b) {
        u64 c[N + 1];
        for (int i = 0; i <= N; ++i) c[i] = 0;
        for (size_t i = 0; i < N; ++i) {
            u64 carry = 0;
            for (int j = 0; j < N; ++j) {
                u128 r = (u128)c[(i + j) % (N + 1)] + (u128)a[i] * b[j] + carry;
                c[(i + j) % (N + 1)] = (u64)r;
                carry = (u64)(r >> 64);
            }
            c[(i + N) % (N + 1)] += carry;
            u64 q = P::MU * c[i];
            carry = 0;
            for (int j = 0; j < N; ++j) {
                u128 r = (u128)c[(i + j) % (N + 1)] + (u128)q * P::ORDER[j] + carry;
                c[(i + j) % (N + 1)] = (u64)r;
                carry = (u64)(r >> 64);
            }
            c[(i + N) % (N + 1)] += carry;
        }
        // After N rounds the window starts at index N: result = [c[N], c[0], ..., c[N-2]]
        Limbs<N> res;
        for (size_t i = 0; i < N; ++i) res[i] = c[(N + i) % (N + 1)];
        if (cmp(res, order()) >= 0) res = sub(res, order());
        return res;
    }

    // monty.rs:110-160 (dedicated square; only instantiated for 4 limbs with p < 2^255)
    static Limbs<N> monty_square(const Limbs<N>& a) {
        Limbs<N> c{};
        u64 hi = 0;
        for (size_t i = 0; i < N; ++i) {
            Limbs<N> u{};
            u64 hi_in = 0;
            for (int j = i + 1; j < N; ++j) {
                u128 t = (u128)a[j] * a[i] + hi_in;
                u[j - (i + 1)] = (u64)t;
                hi_in = (u64)(t >> 64);
            }
            u[N - (i + 1)] = hi_in;
            u = mul2(u);
            u128 t0 = (u128)a[i] * a[i] + c[i];
            c[i] = (u64)t0;
            u64 cin = (u64)(t0 >> 64);
            for (int j = i + 1; j < N; ++j) {
                u128 t = (u128)c[j] + cin + u[j - (i + 1)];
                c[j] = (u64)t;
                cin = (u64)(t >> 64);
            }
            {
                u128 t = (u128)hi + cin + u[N - (i + 1)];
                hi = (u64)t;
            }
            u64 m = c[0] * P::MU;
            u128 t1 = (u128)P::ORDER[0] * m + c[0];
            u64 h = (u64)(t1 >> 64);
            for (int j = 1; j < N; ++j) {
                u128 t = (u128)P::ORDER[j] * m + c[j] + h;
                c[j - 1] = (u64)t;
                h = (u64)(t >> 64);
            }
            u128 t2 = (u128)hi + h;
            c[N - 1] = (u64)t2;
            hi = (u64)(t2 >> 64);
        }
        if (cmp(c, order()) >= 0) c = sub(c, order());
        return c;
    }

    // monty.rs:38-46
    Fp operator+(const Fp& r) const {
        Limbs<N> s = add_no_overflow(limbs, r.limbs);
        Fp o;
        o.limbs = cmp(s, order()) < 0 ? s : sub(s, order());
        return o;
    }
    // monty.rs:58-64
    Fp operator-() const {
        Fp o;
        if (is_zero()) o.limbs = limbs; else o.limbs = sub(order(), limbs);
        return o;
    }
    // monty.rs:48-56
    Fp operator-(const Fp& r) const {
        Fp o;
        if (cmp(limbs, r.limbs) < 0) o.limbs = add_no_overflow(limbs, (-r).limbs);
        else o.limbs = sub(limbs, r.limbs);
        return o;
    }
    Fp operator*(const Fp& r) const { Fp o; o.limbs = monty_multiply(limbs, r.limbs); return o; }
    Fp square() const {
        Fp o;
        if constexpr (P::MONTY_SQUARE) o.limbs = monty_square(limbs);
        else o.limbs = monty_multiply(limbs, limbs);
        return o;
    }
    // field.rs:180-188 (defaults: full multiplies) ; bls12_377_base.rs:229-253 (shift versions)
    Fp dbl() const {
        if constexpr (P::SHIFT_DOUBLE) {
            Limbs<N> r = mul2(limbs);
            Fp o;
            o.limbs = cmp(r, order()) < 0 ? r : sub(r, order());
            return o;
        } else {
            return *this * two();
        }
    }
    Fp triple() const {
        if constexpr (P::SHIFT_DOUBLE) {
            Limbs<N> s = add_no_overflow(mul2(limbs), limbs);
            Limbs<N> ox2 = mul2(order());
            Fp o;
            if (cmp(s, order()) < 0) o.limbs = s;
            else if (cmp(s, ox2) < 0) o.limbs = sub(s, order());
            else o.limbs = sub(s, ox2);
            return o;
        } else {
            return *this * three();
        }
    }
    Fp cube() const { return square() * *this; }

    // monty.rs:169-177 (names inverted in the reference: from_monty = canonical -> Montgomery)
    static Fp from_canonical(const Limbs<N>& c) { Fp o; o.limbs = monty_multiply(c, L<P>(P::R2)); return o; }
    Limbs<N> to_canonical() const {
        Limbs<N> one{};
        one[0] = 1;
        return monty_multiply(limbs, one);
    }
    static Fp from_canonical_u64(u64 v) {
        Limbs<N> c{};
        c[0] = v;
        return from_canonical(c);
    }
    // monty.rs:162-166 ; field.rs:160-166 (zero has no inverse; callers guarantee non-zero)
    Fp inverse_assuming_nonzero() const {
        Fp o;
        o.limbs = monty_multiply(nonzero_multiplicative_inverse(limbs, order()), L<P>(P::R3));
        return o;
    }
    int num_bits_canonical() const {  // field.rs:136-143 num_bits
        Limbs<N> c = to_canonical();
        for (int i = N - 1; i >= 0; --i)
            if (c[i]) return 64 * i + (64 - __builtin_clzll(c[i]));
        return 0;
    }
    // field.rs:309-330
    Fp exp(const Fp& power) const {
        int power_bits = power.num_bits_canonical();
        Fp current = *this, product = one();
        Limbs<N> pc = power.to_canonical();
        for (int l = 0; l < N; ++l) {
            int lim = std::min(64, power_bits);
            for (int j = 0; j < lim; ++j) {
                if ((pc[l] >> j) & 1) product = product * current;
                current = current.square();
            }
            if (power_bits >= 64) power_bits -= 64; else break;
        }
        return product;
    }
    // field.rs:429-435
    static Fp primitive_root_of_unity(int n_power) {
        Fp gen, t;
        gen.limbs = L<P>(P::GENERATOR);
        t.limbs = L<P>(P::T);
        Fp base_root = gen.exp(t);
        return base_root.exp(from_canonical_u64(1ull << (P::TWO_ADICITY - n_power)));
    }
    // field.rs:292-300
    static std::vector<Fp> cyclic_subgroup_known_order(const Fp& g, size_t order_) {
        std::vector<Fp> s;
        s.reserve(order_);
        Fp cur = one();
        for (size_t i = 0; i < order_; ++i) {
            s.push_back(cur);
            cur = cur * g;
        }
        return s;
    }
    // field.rs:251-278 (Montgomery's trick; every element must be non-zero)
    static std::vector<Fp> batch_multiplicative_inverse(const std::vector<Fp>& x) {
        size_t n = x.size();
        if (n == 0) return {};
        std::vector<Fp> a(n);
        a[0] = x[0];
        for (size_t i = 1; i < n; ++i) a[i] = a[i - 1] * x[i];
        std::vector<Fp> a_inv(n);
        a_inv[n - 1] = a[n - 1].inverse_assuming_nonzero();
        for (size_t i = n - 1; i-- > 0;) a_inv[i] = x[i + 1] * a_inv[i + 1];
        std::vector<Fp> x_inv(n);
        x_inv[0] = a_inv[0];
        for (size_t i = 1; i < n; ++i) x_inv[i] = a[i - 1] * a_inv[i];
        return x_inv;
    }
};

// =============================================================================================
// A tiny fork-join pool standing in for Rayon's work-stealing pool.
// =============================================================================================
// Persistent workers (Rayon keeps its pool alive across calls; forking fresh threads per NTT layer would charge the baseline
// twenty thread start-ups per transform).  One job at a time (callers are serialised by job_mu); workers take task indices from a
// shared counter, the calling thread works too.
class WorkerPool {
public:
    static WorkerPool& get() {
        static WorkerPool p;
        return p;
    }
    void run(size_t n_tasks, int threads, const std::function<void(size_t)>& fn) {
        std::lock_guard<std::mutex> job_lock(job_mu_);
        const int helpers = (int)std::min<size_t>((size_t)threads - 1, n_tasks - 1);
        ensure_workers(helpers);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_tasks_ = n_tasks;
            next_.store(0);
            active_ = helpers;
            wanted_ = helpers;
            ++generation_;
        }
        cv_start_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return active_ == 0; });
        fn_ = nullptr;
    }

private:
    WorkerPool() = default;
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            ++generation_;
        }
        cv_start_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void work() {
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= n_tasks_) break;
            (*fn_)(i);
        }
    }
    void ensure_workers(int helpers) {
        while ((int)workers_.size() < helpers) {
            const int id = (int)workers_.size();
            uint64_t born;
            {
                std::lock_guard<std::mutex> lk(mu_);
                born = generation_;  // jobs published before this worker existed are not its business
            }
            workers_.emplace_back([this, id, born] {
                uint64_t seen = born;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_start_.wait(lk, [&] { return generation_ != seen; });
                        seen = generation_;
                        if (stop_) return;
                        if (id >= wanted_) continue;  // this job uses fewer helpers
                    }
                    work();
                    std::lock_guard<std::mutex> lk(mu_);
                    if (--active_ == 0) cv_done_.notify_one();
                }
            });
        }
    }
    std::mutex job_mu_, mu_;
    std::condition_variable cv_start_, cv_done_;
    std::vector<std::thread> workers_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_tasks_ = 0;
    std::atomic<size_t> next_{0};
    int active_ = 0, wanted_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

static thread_local bool t_in_parallel_for = false;
static void parallel_for(size_t n_tasks, int threads, const std::function<void(size_t)>& fn) {
    if (threads <= 1 || n_tasks <= 1 || t_in_parallel_for) {  // a nested loop runs on the thread that reached it
        for (size_t i = 0; i < n_tasks; ++i) fn(i);
        return;
    }
    const std::function<void(size_t)> guarded = [&](size_t i) {
        const bool was = t_in_parallel_for;
        t_in_parallel_for = true;
        fn(i);
        t_in_parallel_for = was;
    };
    WorkerPool::get().run(n_tasks, threads, guarded);
}

// =============================================================================================
// NTT -- src/fft.rs, src/util.rs
// =============================================================================================
static inline unsigned log2_ceil(size_t n) {  // util.rs:2-9
    unsigned r = 0;
    while ((1ull << r) < n) ++r;
    return r;
}
static inline size_t reverse_bits(size_t n, unsigned num_bits) {  // fft.rs:19-26
    size_t r = 0;
    for (unsigned i = 0; i < num_bits; ++i) r |= ((n >> i) & 1) << (num_bits - i - 1);
    return r;
}
template <class T> static std::vector<T> reverse_index_bits(const std::vector<T>& arr) {  // fft.rs:8-17
    size_t n = arr.size();
    unsigned n_power = log2_ceil(n);
    std::vector<T> r;
    r.reserve(n);
    for (size_t i = 0; i < n; ++i) r.push_back(arr[reverse_bits(i, n_power)]);
    return r;
}

template <class F> struct FftPrecomputation {  // fft.rs:28-40
    std::vector<std::vector<F>> subgroups_rev;
    size_t size() const { return subgroups_rev.back().size(); }
};

template <class F> static FftPrecomputation<F> fft_precompute(size_t degree) {  // fft.rs:47-59
    unsigned degree_pow = log2_ceil(degree);
    FftPrecomputation<F> pre;
    for (unsigned i = 0; i <= degree_pow; ++i) {
        F g_i = F::primitive_root_of_unity(i);
        auto subgroup = F::cyclic_subgroup_known_order(g_i, (size_t)1 << i);
        pre.subgroups_rev.push_back(reverse_index_bits(subgroup));
    }
    return pre;
}

// fft.rs:103-156.  Same layer structure, same 2000-pair work chunks (:130), one fork-join per layer.
template <class F>
static std::vector<F> fft_with_precomputation_power_of_2(const std::vector<F>& coefficients,
                                                         const FftPrecomputation<F>& pre, int threads) {
    size_t degree = coefficients.size();
    size_t half_degree = degree >> 1;
    unsigned degree_pow = log2_ceil(degree);
    std::vector<F> evaluations = reverse_index_bits(coefficients);
    const size_t CHUNK = 2000;
    for (unsigned i = 1; i <= degree_pow; ++i) {
        size_t points_per_poly = (size_t)1 << i;
        size_t pairs_per_poly = (size_t)1 << (i - 1);
        std::vector<F> next(degree);
        size_t n_chunks = (half_degree + CHUNK - 1) / CHUNK;
        const std::vector<F>& tw = pre.subgroups_rev[i];
        parallel_for(n_chunks, threads, [&](size_t c) {
            size_t lo = c * CHUNK, hi = std::min(half_degree, lo + CHUNK);
            for (size_t pair_index = lo; pair_index < hi; ++pair_index) {
                size_t poly_index = pair_index / pairs_per_poly;
                size_t within = pair_index % pairs_per_poly;
                size_t child0 = poly_index * points_per_poly + within;
                size_t child1 = child0 + pairs_per_poly;
                F even = evaluations[child0];
                F odd = evaluations[child1];
                F product = tw[within * 2] * odd;
                next[2 * pair_index] = even + product;
                next[2 * pair_index + 1] = even - product;
            }
        });
        evaluations.swap(next);
    }
    return reverse_index_bits(evaluations);
}

// fft.rs:61-80
template <class F>
static std::vector<F> fft_with_precomputation(const std::vector<F>& coefficients, const FftPrecomputation<F>& pre, int threads) {
    size_t degree = coefficients.size();
    size_t padded = (size_t)1 << log2_ceil(degree);
    if (degree == padded) return fft_with_precomputation_power_of_2(coefficients, pre, threads);
    std::vector<F> c = coefficients;
    c.resize(padded, F::zero());
    return fft_with_precomputation_power_of_2(c, pre, threads);
}

// fft.rs:82-101
template <class F>
static std::vector<F> ifft_with_precomputation_power_of_2(const std::vector<F>& points, const FftPrecomputation<F>& pre, int threads) {
    size_t n = points.size();
    F n_inv = F::from_canonical_u64((u64)n).inverse_assuming_nonzero();
    std::vector<F> result = fft_with_precomputation_power_of_2(points, pre, threads);
    result[0] = result[0] * n_inv;
    result[n / 2] = result[n / 2] * n_inv;
    for (size_t i = 1; i < n / 2; ++i) {
        size_t j = n - i;
        F ri = result[j] * n_inv;
        F rj = result[i] * n_inv;
        result[i] = ri;
        result[j] = rj;
    }
    return result;
}

// =============================================================================================
// Polynomial callers of the NTT -- src/polynomial.rs, src/plonk_util.rs
// =============================================================================================
template <class F> static bool poly_is_zero(const std::vector<F>& a) {  // polynomial.rs:88-90
    for (const F& x : a)
        if (!x.is_zero()) return false;
    return true;
}
template <class F> static size_t poly_degree_plus_one(const std::vector<F>& a) {  // polynomial.rs:108-113
    for (size_t i = a.size(); i-- > 0;)
        if (!a[i].is_zero()) return i + 1;
    return 0;
}
template <class F> static void poly_trim(std::vector<F>& a) { a.resize(poly_degree_plus_one(a)); }  // :178-180
template <class F> static void poly_pad(std::vector<F>& a, size_t len) {  // :194-198 (asserts trimmed len <= len)
    poly_trim(a);
    if (a.size() < len) a.resize(len, F::zero());
}
// polynomial.rs:135-143
template <class F> static std::vector<F> poly_eval_domain(const std::vector<F>& a, const FftPrecomputation<F>& pre, int threads) {
    size_t domain_size = pre.size();
    if (a.size() < domain_size) {
        std::vector<F> p = a;
        poly_pad(p, domain_size);
        return fft_with_precomputation(p, pre, threads);
    }
    return fft_with_precomputation(a, pre, threads);
}
// polynomial.rs:330-380
template <class F> static std::vector<F> poly_divide_by_z_h(const std::vector<F>& self, size_t n, int threads) {
    if (poly_is_zero(self)) return self;
    std::vector<F> a_trim = self;
    poly_trim(a_trim);
    F g;
    g.limbs = L<typename F::Params>(F::Params::GENERATOR);
    F g_pow = F::one();
    for (F& x : a_trim) {
        x = x * g_pow;
        g_pow = g * g_pow;
    }
    size_t d = poly_degree_plus_one(a_trim) - 1;
    F root = F::primitive_root_of_unity((int)log2_ceil(d + 1));
    FftPrecomputation<F> pre = fft_precompute<F>(d + 1);
    std::vector<F> a_eval = poly_eval_domain(a_trim, pre, threads);
    F denominator_g = g.exp(F::from_canonical_u64((u64)n));
    F root_n = root.exp(F::from_canonical_u64((u64)n));
    F root_pow = F::one();
    std::vector<F> denominators(a_eval.size());
    for (size_t i = 0; i < a_eval.size(); ++i) {
        if (i != 0) root_pow = root_pow * root_n;
        denominators[i] = denominator_g * root_pow - F::one();
    }
    std::vector<F> denominators_inv = F::batch_multiplicative_inverse(denominators);
    for (size_t i = 0; i < a_eval.size(); ++i) a_eval[i] = a_eval[i] * denominators_inv[i];
    std::vector<F> p = ifft_with_precomputation_power_of_2(a_eval, pre, threads);
    F g_inv = g.inverse_assuming_nonzero();
    F g_inv_pow = F::one();
    for (F& x : p) {
        x = x * g_inv_pow;
        g_inv_pow = g_inv_pow * g_inv;
    }
    return p;
}
// polynomial.rs:208-226
template <class F> static std::vector<F> poly_mul(const std::vector<F>& a, const std::vector<F>& b, int threads) {
    if (poly_is_zero(a) || poly_is_zero(b)) return std::vector<F>(1, F::zero());
    size_t a_deg = poly_degree_plus_one(a) - 1, b_deg = poly_degree_plus_one(b) - 1;
    std::vector<F> a_pad = a, b_pad = b;
    poly_pad(a_pad, a_deg + b_deg + 1);
    poly_pad(b_pad, a_deg + b_deg + 1);
    FftPrecomputation<F> pre = fft_precompute<F>(a_deg + b_deg + 1);
    std::vector<F> a_evals = fft_with_precomputation(a_pad, pre, threads);
    std::vector<F> b_evals = fft_with_precomputation(b_pad, pre, threads);
    std::vector<F> m(a_evals.size());
    for (size_t i = 0; i < m.size(); ++i) m[i] = a_evals[i] * b_evals[i];
    return ifft_with_precomputation_power_of_2(m, pre, threads);
}
// plonk_util.rs:179-190, one polynomial: padded(len * 8) then eval_domain
template <class F> static std::vector<F> poly_to_values_padded(const std::vector<F>& poly, const FftPrecomputation<F>& pre, int threads) {
    std::vector<F> padded = poly;
    poly_pad(padded, poly.size() * 8);
    return poly_eval_domain(padded, pre, threads);
}

// =============================================================================================
// Curves -- src/curve/curve.rs, curve_adds.rs, curve_summations.rs, curve_msm.rs,
//           curve_multiplication.rs.  All in-scope curves have A = 0 (tweedledee_curve.rs:11,
//           tweedledum_curve.rs:11, bls12_377_curve.rs:14); the A != 0 branches of the
//           reference (curve.rs:243, curve_summations.rs:118) are therefore never taken.
// =============================================================================================
template <class BaseP_, class ScalarP_> struct CurveT {
    typedef Fp<BaseP_> Base;
    typedef Fp<ScalarP_> Scalar;
};

template <class C> struct AffinePoint {
    typename C::Base x, y;
    bool zero;
    static AffinePoint ZERO() { return {C::Base::zero(), C::Base::zero(), true}; }
    AffinePoint neg() const { return {x, -y, zero}; }
    bool operator==(const AffinePoint& o) const {  // curve.rs:153-170
        if (zero || o.zero) return zero == o.zero;
        return x == o.x && y == o.y;
    }
};

template <class C> struct ProjectivePoint {
    typedef typename C::Base B;
    B x, y, z;
    bool zero;
    static ProjectivePoint ZERO() { return {B::zero(), B::zero(), B::zero(), true}; }

    // curve.rs:206-214
    AffinePoint<C> to_affine() const {
        if (zero) return AffinePoint<C>::ZERO();
        B z_inv = z.inverse_assuming_nonzero();
        return {x * z_inv, y * z_inv, false};
    }
    // curve.rs:234-260 (A == 0 path)
    ProjectivePoint dbl() const {
        if (zero) return ZERO();
        B xx = x.square();
        B w = xx.triple();
        B s = y.dbl() * z;
        B r = y * s;
        B rr = r.square();
        B b = (x + r).square() - (xx + rr);
        B h = w.square() - b.dbl();
        B x3 = h * s;
        B y3 = w * (b - h) - rr.dbl();
        B z3 = s.cube();
        return {x3, y3, z3, false};
    }
    // curve.rs:280-302
    bool operator==(const ProjectivePoint& o) const {
        if (zero || o.zero) return zero == o.zero;
        return x * o.z == o.x * z && y * o.z == o.y * z;
    }
};

template <class C> static ProjectivePoint<C> to_projective(const AffinePoint<C>& a) {  // curve.rs:93-101
    return {a.x, a.y, C::Base::one(), a.zero};
}

// curve.rs:216-232 (batch_to_affine through batch_multiplicative_inverse_opt, field.rs:223-249)
template <class C> static std::vector<AffinePoint<C>> batch_to_affine(const std::vector<ProjectivePoint<C>>& pp) {
    typedef typename C::Base B;
    std::vector<B> nz;
    nz.reserve(pp.size());
    for (auto& p : pp)
        if (!p.z.is_zero()) nz.push_back(p.z);
    std::vector<B> inv = B::batch_multiplicative_inverse(nz);
    std::vector<AffinePoint<C>> out;
    out.reserve(pp.size());
    size_t k = 0;
    for (auto& p : pp) {
        bool has_inv = !p.z.is_zero();
        B zi = has_inv ? inv[k++] : B::zero();
        if (p.zero) out.push_back(AffinePoint<C>::ZERO());
        else out.push_back({p.x * zi, p.y * zi, false});
    }
    return out;
}

// curve_adds.rs:5-48
template <class C> static ProjectivePoint<C> add_pp(const ProjectivePoint<C>& a, const ProjectivePoint<C>& b) {
    typedef typename C::Base B;
    if (a.zero) return b;
    if (b.zero) return a;
    B x1z2 = a.x * b.z, y1z2 = a.y * b.z, x2z1 = b.x * a.z, y2z1 = b.y * a.z;
    if (x1z2 == x2z1) {
        if (y1z2 == y2z1) return a.dbl();
        if (y1z2 == -y2z1) return ProjectivePoint<C>::ZERO();
    }
    B z1z2 = a.z * b.z;
    B u = y2z1 - y1z2;
    B uu = u.square();
    B v = x2z1 - x1z2;
    B vv = v.square();
    B vvv = v * vv;
    B r = vv * x1z2;
    B aa = uu * z1z2 - vvv - r.dbl();
    return {v * aa, u * (r - aa) - vvv * y1z2, vvv * z1z2, false};
}
// curve_adds.rs:50-90
template <class C> static ProjectivePoint<C> add_pa(const ProjectivePoint<C>& a, const AffinePoint<C>& b) {
    typedef typename C::Base B;
    if (a.zero) return to_projective(b);
    if (b.zero) return a;
    B x2z1 = b.x * a.z, y2z1 = b.y * a.z;
    if (a.x == x2z1) {
        if (a.y == y2z1) return a.dbl();
        if (a.y == -y2z1) return ProjectivePoint<C>::ZERO();
    }
    B u = y2z1 - a.y;
    B uu = u.square();
    B v = x2z1 - a.x;
    B vv = v.square();
    B vvv = v * vv;
    B r = vv * a.x;
    B aa = uu * a.z - vvv - r.dbl();
    return {v * aa, u * (r - aa) - vvv * a.y, vvv * a.z, false};
}
// curve_adds.rs:92-128
template <class C> static ProjectivePoint<C> add_aa(const AffinePoint<C>& a, const AffinePoint<C>& b) {
    typedef typename C::Base B;
    if (a.zero) return to_projective(b);
    if (b.zero) return to_projective(a);
    if (a.x == b.x) {
        if (a.y == b.y) return to_projective(a).dbl();
        if (a.y == -b.y) return ProjectivePoint<C>::ZERO();
    }
    B u = b.y - a.y;
    B uu = u.square();
    B v = b.x - a.x;
    B vv = v.square();
    B vvv = v * vv;
    B r = vv * a.x;
    B aa = uu - vvv - r.dbl();
    return {v * aa, u * (r - aa) - vvv * a.y, vvv, false};
}

// ---- curve_summations.rs ----
template <class C> static std::vector<ProjectivePoint<C>> affine_multisummation_best(std::vector<std::vector<AffinePoint<C>>> s);

// curve_summations.rs:46-58
template <class C> static ProjectivePoint<C> affine_summation_pairwise(const std::vector<AffinePoint<C>>& pts) {
    std::vector<ProjectivePoint<C>> red;
    for (size_t i = 0; i < pts.size(); i += 2) {
        if (i + 1 < pts.size()) red.push_back(add_aa(pts[i], pts[i + 1]));
        else red.push_back(to_projective(pts[i]));
    }
    ProjectivePoint<C> sum = ProjectivePoint<C>::ZERO();
    for (auto& p : red) sum = add_pp(sum, p);
    return sum;
}
// curve_summations.rs:39-43
template <class C> static std::vector<ProjectivePoint<C>> affine_multisummation_pairwise(const std::vector<std::vector<AffinePoint<C>>>& s) {
    std::vector<ProjectivePoint<C>> out;
    out.reserve(s.size());
    for (auto& v : s) out.push_back(affine_summation_pairwise(v));
    return out;
}
// curve_summations.rs:70-158
template <class C> static std::vector<ProjectivePoint<C>> affine_multisummation_batch_inversion(std::vector<std::vector<AffinePoint<C>>> summations) {
    typedef typename C::Base B;
    std::vector<B> to_invert;
    for (auto& s : summations) {
        size_t n = s.size();
        size_t range_end = n == 0 ? 0 : n - 1;
        for (size_t i = 0; i < range_end; i += 2) {
            const AffinePoint<C>&p1 = s[i], &p2 = s[i + 1];
            if (p1.zero || p2.zero || p1 == p2.neg()) {
            } else if (p1 == p2) {
                to_invert.push_back(p1.y.dbl());
            } else {
                to_invert.push_back(p1.x - p2.x);
            }
        }
    }
    std::vector<B> inverses = B::batch_multiplicative_inverse(to_invert);
    std::vector<std::vector<AffinePoint<C>>> all_reduced;
    all_reduced.reserve(summations.size());
    size_t inverse_index = 0;
    for (auto& s : summations) {
        size_t n = s.size();
        std::vector<AffinePoint<C>> red;
        red.reserve((n + 1) / 2);
        size_t range_end = n == 0 ? 0 : n - 1;
        for (size_t i = 0; i < range_end; i += 2) {
            const AffinePoint<C>&p1 = s[i], &p2 = s[i + 1];
            AffinePoint<C> sum;
            if (p1.zero) sum = p2;
            else if (p2.zero) sum = p1;
            else if (p1 == p2.neg()) sum = AffinePoint<C>::ZERO();
            else {
                B inverse = inverses[inverse_index++];
                if (p1 == p2) {
                    B numerator = p1.x.square().triple();
                    B q = numerator * inverse;
                    B x3 = q.square() - p1.x.dbl();
                    B y3 = q * (p1.x - x3) - p1.y;
                    sum = {x3, y3, false};
                } else {
                    B q = (p1.y - p2.y) * inverse;
                    B x3 = q.square() - p1.x - p2.x;
                    B y3 = q * (p1.x - x3) - p1.y;
                    sum = {x3, y3, false};
                }
            }
            red.push_back(sum);
        }
        if (n % 2 == 1) red.push_back(s[n - 1]);
        all_reduced.push_back(std::move(red));
    }
    return affine_multisummation_best<C>(std::move(all_reduced));
}
// curve_summations.rs:24-35
template <class C> static std::vector<ProjectivePoint<C>> affine_multisummation_best(std::vector<std::vector<AffinePoint<C>>> s) {
    size_t pairwise_sums = 0;
    for (auto& v : s) pairwise_sums += v.size() / 2;
    if (pairwise_sums < 70) return affine_multisummation_pairwise<C>(s);
    return affine_multisummation_batch_inversion<C>(std::move(s));
}

// ---- curve_msm.rs ----
template <class C> struct MsmPrecomputation {  // curve_msm.rs:16-25
    std::vector<std::vector<AffinePoint<C>>> powers_per_generator;
    unsigned w;
};
// curve_msm.rs:40-52
template <class C> static std::vector<AffinePoint<C>> precompute_single_generator(const ProjectivePoint<C>& g, unsigned w) {
    unsigned digits = (unsigned)((C::SCALAR_BITS + w - 1) / w);
    std::vector<ProjectivePoint<C>> powers;
    powers.reserve(digits);
    powers.push_back(g);
    for (unsigned i = 1; i < digits; ++i) {
        ProjectivePoint<C> p = powers[i - 1];
        for (unsigned j = 0; j < w; ++j) p = p.dbl();
        powers.push_back(p);
    }
    return batch_to_affine(powers);
}
// curve_msm.rs:27-38
template <class C> static MsmPrecomputation<C> msm_precompute(const std::vector<ProjectivePoint<C>>& gens, unsigned w, int threads) {
    MsmPrecomputation<C> pre;
    pre.w = w;
    pre.powers_per_generator.resize(gens.size());
    const size_t GRAIN = 64;
    size_t n_tasks = (gens.size() + GRAIN - 1) / GRAIN;
    parallel_for(n_tasks, threads, [&](size_t t) {
        size_t lo = t * GRAIN, hi = std::min(gens.size(), lo + GRAIN);
        for (size_t i = lo; i < hi; ++i) pre.powers_per_generator[i] = precompute_single_generator(gens[i], w);
    });
    return pre;
}
// curve_msm.rs:159-180
template <class C> static std::vector<size_t> to_digits(const typename C::Scalar& x, unsigned w) {
    const unsigned scalar_bits = C::SCALAR_BITS;
    unsigned num_digits = (scalar_bits + w - 1) / w;
    auto xc = x.to_canonical();
    std::vector<bool> bits(scalar_bits);
    for (unsigned i = 0; i < scalar_bits; ++i) bits[i] = ((xc[i / 64] >> (i % 64)) & 1) != 0;
    std::vector<size_t> digits;
    digits.reserve(num_digits);
    for (unsigned i = 0; i < num_digits; ++i) {
        size_t d = 0;
        unsigned hi = std::min((i + 1) * w, scalar_bits);
        for (unsigned j = hi; j-- > i * w;) d = (d << 1) | (size_t)bits[j];
        digits.push_back(d);
    }
    return digits;
}
// curve_msm.rs:63-100 (serial)
template <class C> static ProjectivePoint<C> msm_execute(const MsmPrecomputation<C>& pre, const std::vector<typename C::Scalar>& scalars) {
    unsigned w = pre.w;
    size_t base = (size_t)1 << w;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> occ(base);
    for (size_t i = 0; i < scalars.size(); ++i) {
        auto d = to_digits<C>(scalars[i], w);
        for (size_t j = 0; j < d.size(); ++j) occ[d[j]].push_back({(uint32_t)i, (uint32_t)j});
    }
    ProjectivePoint<C> y = ProjectivePoint<C>::ZERO(), u = ProjectivePoint<C>::ZERO();
    for (size_t digit = base - 1; digit >= 1; --digit) {
        for (auto& ij : occ[digit]) u = add_pa(u, pre.powers_per_generator[ij.first][ij.second]);
        y = add_pp(y, u);
    }
    return y;
}
// curve_msm.rs:102-157 (parallel): serial scatter, DIGITS_PER_CHUNK = 80 (:14) parallel chunks, serial tail
template <class C> static ProjectivePoint<C> msm_execute_parallel(const MsmPrecomputation<C>& pre, const std::vector<typename C::Scalar>& scalars, int threads) {
    unsigned w = pre.w;
    size_t base = (size_t)1 << w;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> occ(base);
    for (size_t i = 0; i < scalars.size(); ++i) {
        auto d = to_digits<C>(scalars[i], w);
        for (size_t j = 0; j < d.size(); ++j) occ[d[j]].push_back({(uint32_t)i, (uint32_t)j});
    }
    const size_t DIGITS_PER_CHUNK = 80;
    std::vector<ProjectivePoint<C>> digit_acc(base);
    size_t n_chunks = (base + DIGITS_PER_CHUNK - 1) / DIGITS_PER_CHUNK;
    parallel_for(n_chunks, threads, [&](size_t c) {
        size_t lo = c * DIGITS_PER_CHUNK, hi = std::min(base, lo + DIGITS_PER_CHUNK);
        std::vector<std::vector<AffinePoint<C>>> summations;
        summations.reserve(hi - lo);
        for (size_t digit = lo; digit < hi; ++digit) {
            std::vector<AffinePoint<C>> v;
            v.reserve(occ[digit].size());
            for (auto& ij : occ[digit]) v.push_back(pre.powers_per_generator[ij.first][ij.second]);
            summations.push_back(std::move(v));
        }
        auto res = affine_multisummation_best<C>(std::move(summations));
        for (size_t digit = lo; digit < hi; ++digit) digit_acc[digit] = res[digit - lo];
    });
    ProjectivePoint<C> y = ProjectivePoint<C>::ZERO(), u = ProjectivePoint<C>::ZERO();
    for (size_t digit = base - 1; digit >= 1; --digit) {
        u = add_pp(u, digit_acc[digit]);
        y = add_pp(y, u);
    }
    return y;
}

// ---- curve_multiplication.rs:5-85 (w = 4 Yao single-scalar multiplication) ----
template <class C> static ProjectivePoint<C> scalar_mul(const typename C::Scalar& s, const ProjectivePoint<C>& p) {
    const unsigned WINDOW_BITS = 4, BASE = 16;
    unsigned num_digits = (C::SCALAR_BITS + WINDOW_BITS - 1) / WINDOW_BITS;
    std::vector<ProjectivePoint<C>> powers_proj;
    powers_proj.push_back(p);
    for (unsigned i = 1; i < num_digits; ++i) {
        ProjectivePoint<C> q = powers_proj[i - 1];
        for (unsigned j = 0; j < WINDOW_BITS; ++j) q = q.dbl();
        powers_proj.push_back(q);
    }
    auto powers = batch_to_affine(powers_proj);
    // curve_multiplication.rs:72-85 to_digits: every limb fully split (may exceed num_digits)
    std::vector<u64> digits;
    auto xc = s.to_canonical();
    for (int l = 0; l < C::Scalar::N; ++l)
        for (unsigned j = 0; j < 64 / WINDOW_BITS; ++j) digits.push_back((xc[l] >> (j * WINDOW_BITS)) % BASE);
    ProjectivePoint<C> y = ProjectivePoint<C>::ZERO(), u = ProjectivePoint<C>::ZERO();
    for (unsigned j = BASE - 1; j >= 1; --j) {
        std::vector<AffinePoint<C>> summands;
        for (size_t i = 0; i < digits.size() && i < powers.size(); ++i)
            if (digits[i] == j) summands.push_back(powers[i]);
        std::vector<std::vector<AffinePoint<C>>> one;
        one.push_back(std::move(summands));
        u = add_pp(u, affine_multisummation_batch_inversion<C>(std::move(one))[0]);
        y = add_pp(y, u);
    }
    return y;
}
// bls12_377_curve.rs:65-85 mul_naive: double-and-add over the canonical bits
template <class C> static ProjectivePoint<C> mul_naive(const typename C::Scalar& s, const ProjectivePoint<C>& p) {
    auto xc = s.to_canonical();
    ProjectivePoint<C> g = p, sum = ProjectivePoint<C>::ZERO();
    for (int l = 0; l < C::Scalar::N; ++l)
        for (int j = 0; j < 64; ++j) {
            if ((xc[l] >> j) & 1) sum = add_pp(sum, g);
            g = g.dbl();
        }
    return sum;
}

// Concrete curves (constants: tweedledee_curve.rs:11-18, tweedledum_curve.rs:11-33, bls12_377_curve.rs:14-33)
struct Tweedledee : CurveT<TweedledeeBaseP, TweedledumBaseP> {
    static constexpr unsigned SCALAR_BITS = 255;
    static AffinePoint<Tweedledee> generator() {
        Base x, y;
        x = -Base::one();       // NEG_ONE
        y = Base::two();        // TWO
        return {x, y, false};
    }
};
struct Tweedledum : CurveT<TweedledumBaseP, TweedledeeBaseP> {
    static constexpr unsigned SCALAR_BITS = 255;
    static AffinePoint<Tweedledum> generator() {
        Base x = Base::one(), y;
        y.limbs = {12815994359195135157ull, 12442237869110527732ull, 9256472484777506843ull, 1114242145010923164ull};
        return {x, y, false};
    }
};
struct Bls12377 : CurveT<Bls12377BaseP, Bls12377ScalarP> {
    static constexpr unsigned SCALAR_BITS = 253;
    static AffinePoint<Bls12377> generator() {
        Base x, y;
        x.limbs = {2742467569752756724ull, 14217256487979144792ull, 6635299530028159197ull, 8509097278468658840ull, 14518893593143693938ull, 46181716169194829ull};
        y.limbs = {9336971515457667571ull, 28021381849722296ull, 18085035374859187530ull, 14013031479170682136ull, 3369780711397861396ull, 35370409237953649ull};
        return {x, y, false};
    }
};

// pallas_curve.rs:7-19, vesta_curve.rs:7-19: y^2 = x^3 + 5, generator (-1, 2)
struct Pallas : CurveT<PallasBaseP, VestaBaseP> {
    static constexpr unsigned SCALAR_BITS = 255;
    static AffinePoint<Pallas> generator() { return {-Base::one(), Base::two(), false}; }
};
struct Vesta : CurveT<VestaBaseP, PallasBaseP> {
    static constexpr unsigned SCALAR_BITS = 255;
    static AffinePoint<Vesta> generator() { return {-Base::one(), Base::two(), false}; }
};

#include "plonk_gates.inc"
#include "serialization.inc"

// =============================================================================================
// C interface for ctypes (tests / cpu_baseline only)
// =============================================================================================
template <class F> static F ld(const u64* p) {
    F f;
    for (int i = 0; i < F::N; ++i) f.limbs[i] = p[i];
    return f;
}
template <class F> static void st(u64* p, const F& f) {
    for (int i = 0; i < F::N; ++i) p[i] = f.limbs[i];
}
template <class C> static AffinePoint<C> ld_aff(const u64* xy, const uint8_t* zero, size_t i) {
    typedef typename C::Base B;
    AffinePoint<C> a;
    a.x = ld<B>(xy + i * 2 * B::N);
    a.y = ld<B>(xy + i * 2 * B::N + B::N);
    a.zero = zero ? zero[i] != 0 : false;
    return a;
}
template <class C> static void st_aff(u64* xy, uint8_t* zero, size_t i, const AffinePoint<C>& a) {
    typedef typename C::Base B;
    st(xy + i * 2 * B::N, a.x);
    st(xy + i * 2 * B::N + B::N, a.y);
    if (zero) zero[i] = a.zero ? 1 : 0;
}

#define FIELD_DISPATCH(field, ...)                  \
    switch (field) {                                \
        case 0: { typedef Fp<TweedledeeBaseP> F; __VA_ARGS__; } break; \
        case 1: { typedef Fp<TweedledumBaseP> F; __VA_ARGS__; } break; \
        case 2: { typedef Fp<Bls12377ScalarP> F; __VA_ARGS__; } break; \
        case 3: { typedef Fp<Bls12377BaseP> F; __VA_ARGS__; } break;   \
        case 4: { typedef Fp<PallasBaseP> F; __VA_ARGS__; } break;     \
        case 5: { typedef Fp<VestaBaseP> F; __VA_ARGS__; } break;      \
        default: return -1;                         \
    }
#define CURVE_DISPATCH(curve, ...)                  \
    switch (curve) {                                \
        case 0: { typedef Tweedledee C; __VA_ARGS__; } break; \
        case 1: { typedef Tweedledum C; __VA_ARGS__; } break; \
        case 2: { typedef Bls12377 C; __VA_ARGS__; } break;   \
        case 3: { typedef Pallas C; __VA_ARGS__; } break;     \
        case 4: { typedef Vesta C; __VA_ARGS__; } break;      \
        default: return -1;                         \
    }

template <class F> static int field_binop_t(int op, const u64* a, const u64* b, u64* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        F x = ld<F>(a + i * F::N), y = ld<F>(b + i * F::N), r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            default: return -1;
        }
        st(out + i * F::N, r);
    }
    return 0;
}
template <class F> static int field_unop_t(int op, const u64* a, u64* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        F x = ld<F>(a + i * F::N), r;
        switch (op) {
            case 0: r = -x; break;
            case 1: r = x.square(); break;
            case 2: r = x.is_zero() ? x : x.inverse_assuming_nonzero(); break;
            case 3: r.limbs = x.to_canonical(); break;
            case 4: r = F::from_canonical(x.limbs); break;
            case 5: r = x.dbl(); break;
            case 6: r = x.triple(); break;
            default: return -1;
        }
        st(out + i * F::N, r);
    }
    return 0;
}

template <class F> struct FftHandle { FftPrecomputation<F> pre; };
template <class C> struct MsmHandle { MsmPrecomputation<C> pre; };
struct AnyHandle { int kind; int id; void* ptr; };

// which: 0 ORDER 1 R 2 R2 3 R3 4 MU(in limb 0) 5 TWO 6 THREE 7 GENERATOR 8 T 9 NEG_ONE
template <class F, class P> static int field_const_t(int which, u64* out) {
    Limbs<P::N> v{};
    switch (which) {
        case 0: v = L<P>(P::ORDER); break;
        case 1: v = L<P>(P::R); break;
        case 2: v = L<P>(P::R2); break;
        case 3: v = L<P>(P::R3); break;
        case 4: v[0] = P::MU; break;
        case 5: v = L<P>(P::TWO); break;
        case 6: v = L<P>(P::THREE); break;
        case 7: v = L<P>(P::GENERATOR); break;
        case 8: v = L<P>(P::T); break;
        case 9: v = (-F::one()).limbs; break;
        default: return -1;
    }
    for (int i = 0; i < P::N; ++i) out[i] = v[i];
    return 0;
}

extern "C" {

int orc_field_limbs(int field) { FIELD_DISPATCH(field, return F::N); return -1; }
int orc_field_binop(int field, int op, const u64* a, const u64* b, u64* out, size_t n) {
    FIELD_DISPATCH(field, return field_binop_t<F>(op, a, b, out, n));
    return -1;
}
int orc_field_unop(int field, int op, const u64* a, u64* out, size_t n) {
    FIELD_DISPATCH(field, return field_unop_t<F>(op, a, out, n));
    return -1;
}
int orc_field_const(int field, int which, u64* out) {
    switch (field) {
        case 0: return field_const_t<Fp<TweedledeeBaseP>, TweedledeeBaseP>(which, out);
        case 1: return field_const_t<Fp<TweedledumBaseP>, TweedledumBaseP>(which, out);
        case 2: return field_const_t<Fp<Bls12377ScalarP>, Bls12377ScalarP>(which, out);
        case 3: return field_const_t<Fp<Bls12377BaseP>, Bls12377BaseP>(which, out);
        case 4: return field_const_t<Fp<PallasBaseP>, PallasBaseP>(which, out);
        case 5: return field_const_t<Fp<VestaBaseP>, VestaBaseP>(which, out);
    }
    return -1;
}
int orc_root_of_unity(int field, int n_power, u64* out) {
    FIELD_DISPATCH(field, { st(out, F::primitive_root_of_unity(n_power)); return 0; });
    return -1;
}
int orc_batch_inverse(int field, const u64* in, u64* out, size_t n) {
    FIELD_DISPATCH(field, {
        std::vector<F> x(n);
        for (size_t i = 0; i < n; ++i) x[i] = ld<F>(in + i * F::N);
        auto r = F::batch_multiplicative_inverse(x);
        for (size_t i = 0; i < n; ++i) st(out + i * F::N, r[i]);
        return 0;
    });
    return -1;
}
int orc_div2(int n_limbs, const u64* in, u64* out) {  // bigint_arithmetic.rs:84-91 (KAT :134-156)
    if (n_limbs == 6) { Limbs<6> x; for (int i = 0; i < 6; ++i) x[i] = in[i]; x = div2(x); for (int i = 0; i < 6; ++i) out[i] = x[i]; return 0; }
    if (n_limbs == 4) { Limbs<4> x; for (int i = 0; i < 4; ++i) x[i] = in[i]; x = div2(x); for (int i = 0; i < 4; ++i) out[i] = x[i]; return 0; }
    return -1;
}
u64 orc_reverse_bits(u64 n, unsigned num_bits) { return reverse_bits(n, num_bits); }

// ---- NTT ----
void* orc_fft_precompute(int field, size_t degree) {
    AnyHandle* h = new AnyHandle{0, field, nullptr};
    switch (field) {
        case 0: h->ptr = new FftHandle<Fp<TweedledeeBaseP>>{fft_precompute<Fp<TweedledeeBaseP>>(degree)}; break;
        case 1: h->ptr = new FftHandle<Fp<TweedledumBaseP>>{fft_precompute<Fp<TweedledumBaseP>>(degree)}; break;
        case 2: h->ptr = new FftHandle<Fp<Bls12377ScalarP>>{fft_precompute<Fp<Bls12377ScalarP>>(degree)}; break;
        case 3: h->ptr = new FftHandle<Fp<Bls12377BaseP>>{fft_precompute<Fp<Bls12377BaseP>>(degree)}; break;
        case 4: h->ptr = new FftHandle<Fp<PallasBaseP>>{fft_precompute<Fp<PallasBaseP>>(degree)}; break;
        case 5: h->ptr = new FftHandle<Fp<VestaBaseP>>{fft_precompute<Fp<VestaBaseP>>(degree)}; break;
        default: delete h; return nullptr;
    }
    return h;
}
int orc_fft_free(void* hv) {
    AnyHandle* h = (AnyHandle*)hv;
    if (!h) return -1;
    FIELD_DISPATCH(h->id, delete (FftHandle<F>*)h->ptr);
    delete h;
    return 0;
}
long orc_fft_table_size(void* hv) {
    AnyHandle* h = (AnyHandle*)hv;
    FIELD_DISPATCH(h->id, return (long)((FftHandle<F>*)h->ptr)->pre.size());
    return -1;
}
int orc_fft_table_layer(void* hv, unsigned layer, u64* out) {
    AnyHandle* h = (AnyHandle*)hv;
    FIELD_DISPATCH(h->id, {
        auto& t = ((FftHandle<F>*)h->ptr)->pre.subgroups_rev;
        if (layer >= t.size()) return -1;
        for (size_t i = 0; i < t[layer].size(); ++i) st(out + i * F::N, t[layer][i]);
        return 0;
    });
    return -1;
}
// mode 0: fft_with_precomputation (pads to pow2; out must hold the padded length)
// mode 1: fft_with_precomputation_power_of_2   mode 2: ifft_with_precomputation_power_of_2
int orc_fft(void* hv, int mode, const u64* in, size_t n, u64* out, int threads) {
    AnyHandle* h = (AnyHandle*)hv;
    FIELD_DISPATCH(h->id, {
        auto& pre = ((FftHandle<F>*)h->ptr)->pre;
        std::vector<F> c(n);
        for (size_t i = 0; i < n; ++i) c[i] = ld<F>(in + i * F::N);
        std::vector<F> r;
        if (mode == 0) r = fft_with_precomputation(c, pre, threads);
        else if (mode == 1) r = fft_with_precomputation_power_of_2(c, pre, threads);
        else if (mode == 2) r = ifft_with_precomputation_power_of_2(c, pre, threads);
        else return -1;
        for (size_t i = 0; i < r.size(); ++i) st(out + i * F::N, r[i]);
        return 0;
    });
    return -1;
}

// ---- polynomial callers ----
// out must hold max(len, 2^ceil(log2(len))) elements; *out_len receives the result length
int orc_poly_divide_by_z_h(int field, const u64* in, size_t len, size_t n, u64* out, size_t* out_len, int threads) {
    FIELD_DISPATCH(field, {
        std::vector<F> a(len);
        for (size_t i = 0; i < len; ++i) a[i] = ld<F>(in + i * F::N);
        std::vector<F> r = poly_divide_by_z_h(a, n, threads);
        for (size_t i = 0; i < r.size(); ++i) st(out + i * F::N, r[i]);
        *out_len = r.size();
        return 0;
    });
    return -1;
}
// out must hold 2^ceil(log2(la + lb)) elements
int orc_poly_mul(int field, const u64* a_in, size_t la, const u64* b_in, size_t lb, u64* out, size_t* out_len, int threads) {
    FIELD_DISPATCH(field, {
        std::vector<F> a(la), b(lb);
        for (size_t i = 0; i < la; ++i) a[i] = ld<F>(a_in + i * F::N);
        for (size_t i = 0; i < lb; ++i) b[i] = ld<F>(b_in + i * F::N);
        std::vector<F> r = poly_mul(a, b, threads);
        for (size_t i = 0; i < r.size(); ++i) st(out + i * F::N, r[i]);
        *out_len = r.size();
        return 0;
    });
    return -1;
}
// polynomials_to_values_padded for one polynomial against an orc_fft_precompute handle; out holds the domain size
int orc_poly_to_values_padded(void* hv, const u64* in, size_t len, u64* out, int threads) {
    AnyHandle* h = (AnyHandle*)hv;
    FIELD_DISPATCH(h->id, {
        auto& pre = ((FftHandle<F>*)h->ptr)->pre;
        if (len * 8 > pre.size()) return -2;  // the reference debug-asserts the table size (fft.rs:107-111)
        std::vector<F> a(len);
        for (size_t i = 0; i < len; ++i) a[i] = ld<F>(in + i * F::N);
        std::vector<F> r = poly_to_values_padded(a, pre, threads);
        for (size_t i = 0; i < r.size(); ++i) st(out + i * F::N, r[i]);
        return 0;
    });
    return -1;
}

// ---- curves ----
int orc_curve_generator(int curve, u64* out_xy) {
    CURVE_DISPATCH(curve, { st_aff<C>(out_xy, nullptr, 0, C::generator()); return 0; });
    return -1;
}
// op 0: a + b (affine+affine -> projective -> affine)   op 1: affine double of a (curve.rs:112-135 equivalent)
// op 2: [s]a via curve_multiplication.rs   op 3: [s]a via mul_naive   (s = Montgomery scalar limbs in b_or_s)
int orc_curve_op(int curve, int op, const u64* a_xy, uint8_t a_zero, const u64* b_or_s, uint8_t b_zero, u64* out_xy, uint8_t* out_zero) {
    CURVE_DISPATCH(curve, {
        AffinePoint<C> a = ld_aff<C>(a_xy, &a_zero, 0);
        AffinePoint<C> r;
        if (op == 0) {
            AffinePoint<C> b = ld_aff<C>(b_or_s, &b_zero, 0);
            r = add_aa(a, b).to_affine();
        } else if (op == 1) {
            r = to_projective(a).dbl().to_affine();
        } else if (op == 2) {
            r = scalar_mul<C>(ld<typename C::Scalar>(b_or_s), to_projective(a)).to_affine();
        } else if (op == 3) {
            r = mul_naive<C>(ld<typename C::Scalar>(b_or_s), to_projective(a)).to_affine();
        } else return -1;
        st_aff<C>(out_xy, out_zero, 0, r);
        return 0;
    });
    return -1;
}
// out_i = [s_i] base for n Montgomery scalars: the reference's CurveScalar * ProjectivePoint (curve_multiplication.rs:5-70, w = 4)
// once per scalar, spread over `threads` workers.  The parity tests use it for generators WITHOUT structure: G_i = [h_i] G for
// seeded h_i (curve_msm.rs:218-241 tests the MSM over arbitrary generators against the sum of scalar multiplications).
int orc_scalar_mul_batch(int curve, size_t n, const u64* scalars, const u64* base_xy, int threads, u64* out_xy, uint8_t* out_zero) {
    CURVE_DISPATCH(curve, {
        uint8_t bz = 0;
        const ProjectivePoint<C> b = to_projective(ld_aff<C>(base_xy, &bz, 0));
        const size_t chunk = 64, n_chunks = (n + chunk - 1) / chunk;
        parallel_for(n_chunks, threads, [&](size_t c) {
            for (size_t i = c * chunk; i < n && i < (c + 1) * chunk; ++i)
                st_aff<C>(out_xy, out_zero, i, scalar_mul<C>(ld<typename C::Scalar>(scalars + i * C::Scalar::N), b).to_affine());
        });
        return 0;
    });
    return -1;
}
// mode 0 pairwise, 1 batch inversion, 2 best  (curve_summations.rs:18-158) -> affine
int orc_affine_summation(int curve, int mode, size_t n, const u64* pts_xy, const uint8_t* zero, u64* out_xy, uint8_t* out_zero) {
    CURVE_DISPATCH(curve, {
        std::vector<AffinePoint<C>> v(n);
        for (size_t i = 0; i < n; ++i) v[i] = ld_aff<C>(pts_xy, zero, i);
        ProjectivePoint<C> r;
        std::vector<std::vector<AffinePoint<C>>> one;
        one.push_back(v);
        if (mode == 0) r = affine_summation_pairwise(v);
        else if (mode == 1) r = affine_multisummation_batch_inversion<C>(std::move(one))[0];
        else r = affine_multisummation_best<C>(std::move(one))[0];
        st_aff<C>(out_xy, out_zero, 0, r.to_affine());
        return 0;
    });
    return -1;
}
int orc_to_digits(int curve, const u64* scalar, unsigned w, u64* out, size_t* n_out) {
    CURVE_DISPATCH(curve, {
        auto d = to_digits<C>(ld<typename C::Scalar>(scalar), w);
        for (size_t i = 0; i < d.size(); ++i) out[i] = d[i];
        *n_out = d.size();
        return 0;
    });
    return -1;
}
// Synthetic bases B_i = G0 + i*D, affine, by n projective+affine additions.
int orc_gen_bases(int curve, size_t n, const u64* g0_xy, const u64* d_xy, u64* out_xy) {
    CURVE_DISPATCH(curve, {
        AffinePoint<C> g0 = ld_aff<C>(g0_xy, nullptr, 0), d = ld_aff<C>(d_xy, nullptr, 0);
        std::vector<ProjectivePoint<C>> pts(n);
        ProjectivePoint<C> cur = to_projective(g0);
        for (size_t i = 0; i < n; ++i) {
            pts[i] = cur;
            cur = add_pa(cur, d);
        }
        auto aff = batch_to_affine(pts);
        for (size_t i = 0; i < n; ++i) st_aff<C>(out_xy, nullptr, i, aff[i]);
        return 0;
    });
    return -1;
}
void* orc_msm_precompute(int curve, size_t n, const u64* bases_xy, const uint8_t* zero, unsigned w, int threads) {
    AnyHandle* h = new AnyHandle{1, curve, nullptr};
    switch (curve) {
#define MK(ID, C)                                                                                 \
    case ID: {                                                                                    \
        std::vector<ProjectivePoint<C>> g(n);                                                     \
        for (size_t i = 0; i < n; ++i) g[i] = to_projective(ld_aff<C>(bases_xy, zero, i));        \
        h->ptr = new MsmHandle<C>{msm_precompute<C>(g, w, threads)};                              \
    } break;
        MK(0, Tweedledee) MK(1, Tweedledum) MK(2, Bls12377) MK(3, Pallas) MK(4, Vesta)
#undef MK
        default: delete h; return nullptr;
    }
    return h;
}
int orc_msm_free(void* hv) {
    AnyHandle* h = (AnyHandle*)hv;
    if (!h) return -1;
    CURVE_DISPATCH(h->id, delete (MsmHandle<C>*)h->ptr);
    delete h;
    return 0;
}
// level j of generator i of the table: [2^(w j)] G_i affine (curve_msm.rs:19-21)
int orc_msm_table_entry(void* hv, size_t i, size_t j, u64* out_xy, uint8_t* out_zero) {
    AnyHandle* h = (AnyHandle*)hv;
    CURVE_DISPATCH(h->id, {
        auto& t = ((MsmHandle<C>*)h->ptr)->pre.powers_per_generator;
        if (i >= t.size() || j >= t[i].size()) return -1;
        st_aff<C>(out_xy, out_zero, 0, t[i][j]);
        return 0;
    });
    return -1;
}
// parallel 0: msm_execute (serial, curve_msm.rs:63)   1: msm_execute_parallel (:102).  Output = to_affine().
// out_proj (optional, 3*L limbs): the raw projective limbs, to show they are order dependent.
int orc_msm_execute(void* hv, const u64* scalars, size_t n, int parallel, int threads, u64* out_xy, uint8_t* out_zero, u64* out_proj) {
    AnyHandle* h = (AnyHandle*)hv;
    CURVE_DISPATCH(h->id, {
        auto& pre = ((MsmHandle<C>*)h->ptr)->pre;
        if (pre.powers_per_generator.size() != n) return -2;  // assert_eq! curve_msm.rs:67,106
        std::vector<typename C::Scalar> s(n);
        for (size_t i = 0; i < n; ++i) s[i] = ld<typename C::Scalar>(scalars + i * C::Scalar::N);
        ProjectivePoint<C> r = parallel ? msm_execute_parallel<C>(pre, s, threads) : msm_execute<C>(pre, s);
        st_aff<C>(out_xy, out_zero, 0, r.to_affine());
        if (out_proj) {
            st(out_proj, r.x);
            st(out_proj + C::Base::N, r.y);
            st(out_proj + 2 * C::Base::N, r.z);
        }
        return 0;
    });
    return -1;
}

// ---- seeded inputs: SplitMix64 + rejection sampling as rand_range_from_rng (bigint_arithmetic.rs:98-117) ----
static inline u64 splitmix64(u64& state) {
    state += 0x9E3779B97F4A7C15ull;
    u64 z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
int orc_rand_field(int field, u64 seed, size_t count, u64* out) {
    FIELD_DISPATCH(field, {
        Limbs<F::N> order = F::order();
        int strip = __builtin_clzll(order[F::N - 1]);
        u64 state = seed;
        for (size_t k = 0; k < count;) {
            Limbs<F::N> l;
            for (int i = 0; i < F::N; ++i) l[i] = splitmix64(state);
            l[F::N - 1] >>= strip;
            if (cmp(l, order) < 0) {
                for (int i = 0; i < F::N; ++i) out[k * F::N + i] = l[i];
                ++k;
            }
        }
        return 0;
    });
    return -1;
}


// ---- Plonk quotient numerator (plonk_gates.inc) ----
// gate < 0: evaluate_all_constraints; else Gate::evaluate_filtered of that gate (order of gates/mod.rs:52-113).  unfiltered != 0: evaluate_unfiltered.
int orc_gate_constraints(int field, int gate, int unfiltered, const u64* k, const u64* l, const u64* r, const u64* b, const u64* zeta, const u64* a,
                         u64* out, size_t* n_out) {
    FIELD_DISPATCH(field, {
        if (F::N != 4) return -1;
        F kv[plonk::NUM_CONSTANTS], lv[plonk::NUM_WIRES], rv[plonk::NUM_WIRES], bv[plonk::NUM_WIRES];
        for (size_t j = 0; j < plonk::NUM_CONSTANTS; ++j) kv[j] = ld<F>(k + j * F::N);
        for (size_t j = 0; j < plonk::NUM_WIRES; ++j) {
            lv[j] = ld<F>(l + j * F::N);
            rv[j] = ld<F>(r + j * F::N);
            bv[j] = ld<F>(b + j * F::N);
        }
        plonk::GateEnv<F> env{kv, lv, rv, bv, ld<F>(zeta), ld<F>(a)};
        std::vector<F> c = gate < 0 ? plonk::evaluate_all_constraints(env) : unfiltered ? plonk::gate_unfiltered(gate, env) : plonk::gate_filtered(gate, env);
        for (size_t i = 0; i < c.size(); ++i) st(out + i * F::N, c[i]);
        *n_out = c.size();
        return 0;
    });
    return -1;
}
int orc_eval_l_1(int field, size_t n, const u64* x, u64* out) {
    FIELD_DISPATCH(field, { st(out, plonk::eval_l_1(n, ld<F>(x))); return 0; });
    return -1;
}
int orc_mds(int field, size_t n, size_t r, size_t c, u64* out) {
    FIELD_DISPATCH(field, { st(out, plonk::mds_get<F>(n, r, c)); return 0; });
    return -1;
}
// plonk.rs:392-453; tables row-major: constants 6 x 8n, wires 9 x 8n, s_sigma 6 x 8n, z 8n, k_is 6
int orc_vanishing_points(int field, size_t degree, const u64* constants, const u64* wires, const u64* s_sigma, const u64* z, const u64* k_is,
                         const u64* alpha, const u64* beta, const u64* gamma, const u64* zeta, const u64* a, u64* out, int threads) {
    FIELD_DISPATCH(field, {
        if (F::N != 4) return -1;
        const size_t n8 = 8 * degree;
        auto load = [&](const u64* p, size_t cnt) {
            std::vector<F> v(cnt);
            for (size_t i = 0; i < cnt; ++i) v[i] = ld<F>(p + i * F::N);
            return v;
        };
        const std::vector<F> cv = load(constants, plonk::NUM_CONSTANTS * n8), wv = load(wires, plonk::NUM_WIRES * n8),
                             sv = load(s_sigma, plonk::NUM_ROUTED_WIRES * n8), zv = load(z, n8), kv = load(k_is, plonk::NUM_ROUTED_WIRES);
        const std::vector<F> res = plonk::vanishing_points<F>(degree, cv.data(), wv.data(), sv.data(), zv.data(), kv.data(), ld<F>(alpha), ld<F>(beta),
                                                              ld<F>(gamma), ld<F>(zeta), ld<F>(a), threads);
        for (size_t i = 0; i < n8; ++i) st(out + i * F::N, res[i]);
        return 0;
    });
    return -1;
}

// ---- canonical byte encodings (serialization.inc) ----
int orc_field_to_bytes(int field, const u64* x, size_t n, uint8_t* out) {
    FIELD_DISPATCH(field, { for (size_t i = 0; i < n; ++i) ser::to_canonical_u8_vec<F>(ld<F>(x + i * F::N), out + i * F::N * 8); return 0; });
    return -1;
}
// returns the number of records that were "Out of range" (their output is left zero)
int orc_field_from_bytes(int field, const uint8_t* in, size_t n, u64* out) {
    FIELD_DISPATCH(field, {
        int bad = 0;
        for (size_t i = 0; i < n; ++i) {
            F v = F::zero();
            if (!ser::from_canonical_u8_vec<F>(in + i * F::N * 8, v)) { ++bad; v = F::zero(); }
            st(out + i * F::N, v);
        }
        return bad;
    });
    return -1;
}
int orc_field_sqrt(int field, const u64* x, u64* out) {  // 1: root written, 0: not a square
    FIELD_DISPATCH(field, { F r; if (!ser::square_root(ld<F>(x), r)) return 0; st(out, r); return 1; });
    return -1;
}
static u64 curve_b_small(int curve) { return curve == 1 ? 7 : curve == 2 ? 1 : 5; }  // + pallas_curve.rs:12, vesta_curve.rs:12 (5)  // tweedledee_curve.rs:12, tweedledum_curve.rs:12-13, bls12_377_curve.rs:15
int orc_point_to_bytes(int curve, const u64* xy, const uint8_t* zero, size_t n, uint8_t* out) {
    CURVE_DISPATCH(curve, {
        const size_t rec = 1 + C::Base::N * 8;
        for (size_t i = 0; i < n; ++i) ser::write_point<C>(ld_aff<C>(xy, zero, i), out + i * rec);
        return 0;
    });
    return -1;
}
int orc_point_from_bytes(int curve, const uint8_t* in, size_t n, u64* out_xy, uint8_t* out_zero, uint8_t* status) {
    CURVE_DISPATCH(curve, {
        const size_t rec = 1 + C::Base::N * 8;
        const typename C::Base b = C::Base::from_canonical_u64(curve_b_small(curve));
        for (size_t i = 0; i < n; ++i) {
            AffinePoint<C> p = AffinePoint<C>::ZERO();
            status[i] = (uint8_t)ser::read_point<C>(in + i * rec, b, p);
            if (status[i]) p = {C::Base::zero(), C::Base::zero(), false};
            st_aff<C>(out_xy, out_zero, i, p);
        }
        return 0;
    });
    return -1;
}
}  // extern "C"
