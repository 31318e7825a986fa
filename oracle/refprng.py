"""oracle/refprng.py -- TEST INFRASTRUCTURE ONLY.

Restatement of the reference's test PRNG (helpers/prng_unsafe.nim), as far as its MSM regression test needs it
(tests/math_elliptic_curves/t_ec_shortw_jac_g2_msm_bug_366.nim:17-43: seed 1234, 22529 scalars from random_long01Seq):

    seed / splitMix64 ................ prng_unsafe.nim:50-69   (NB: the reference's splitMix64 multiplies by 0xbf58476d1ce4e5b9 TWICE --
                                                                 not the textbook second constant -- restated as written)
    next (xoshiro512**) .............. :71-93
    random_unsafe(maxExclusive) ...... :98-123                 (Lemire's bounded integers, O'Neill's variant, on the low 32 bits of next)
    sample_unsafe .................... :134-136
    random_long01Seq(bytes) .......... :233-247                (runs of equal bits, secp256k1's testrand; lengths from two draws)
    random_long01Seq(BigInt) ......... :249-260                (a draw decides big- or little-endian interpretation of the buffer)
    random_long01Seq(FF) ............. :270-276                (a 2N-limb BigInt reduced mod the field's modulus)

Parity of this file is UNPINNED: the reference holds no known-answer vector for its PRNG (the test only asserts that its two MSM
implementations agree on the inputs), and Nim is not available to print one.  It is a line-by-line restatement; what the tests built on
it claim is "the reference's input construction", not "the reference's exact 22529 scalars".
"""

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1


def _rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M64


class RngState:
    def __init__(self, seed=None):
        self.s = [0] * 8
        if seed is not None:
            self.seed(seed)

    def seed(self, x):
        sm = [x & M64]

        def split_mix():
            sm[0] = (sm[0] + 0x9E3779B97F4A7C15) & M64
            r = sm[0]
            r = ((r ^ (r >> 30)) * 0xBF58476D1CE4E5B9) & M64
            r = ((r ^ (r >> 27)) * 0xBF58476D1CE4E5B9) & M64      # (sic: prng_unsafe.nim:57)
            return r ^ (r >> 31)
        self.s = [split_mix() for _ in range(8)]

    def next(self):
        s = self.s
        result = (_rotl((s[1] * 5) & M64, 7) * 9) & M64
        t = (s[1] << 11) & M64
        s[2] ^= s[0]
        s[5] ^= s[1]
        s[1] ^= s[2]
        s[7] ^= s[3]
        s[3] ^= s[4]
        s[4] ^= s[5]
        s[0] ^= s[6]
        s[6] ^= s[7]
        s[6] ^= t
        s[7] = _rotl(s[7], 21)
        return result

    def random_unsafe(self, max_exclusive):
        mx = max_exclusive & M32
        x = self.next() & M32
        m = x * mx
        low = m & M32
        if low < mx:
            t = (-mx) & M32
            if t >= mx:
                t -= mx
                if t >= mx:
                    t %= mx
            while low < t:
                x = self.next() & M32
                m = x * mx
                low = m & M32
        return m >> 32

    def sample_unsafe(self, src):
        return src[self.random_unsafe(len(src))]

    def random_long01seq_bytes(self, nbytes):
        a = bytearray(nbytes)
        bits, bit = nbytes * 8, 0
        while bit < bits:
            now = 1 + ((self.random_unsafe(1 << 6) * self.random_unsafe(1 << 5) + 16) & M32) // 31
            val = self.sample_unsafe([0, 1])
            while now > 0 and bit < bits:
                a[bit >> 3] |= (val << (bit & 7)) & 0xFF
                now -= 1
                bit += 1
        return bytes(a)

    def random_long01seq_bigint(self, bits):
        buf = self.random_long01seq_bytes((bits + 7) // 8)
        order = self.sample_unsafe(["big", "little"])
        return int.from_bytes(buf, order) & ((1 << bits) - 1)     # unmarshal + clearExtraBitsOverMSB

    def random_long01seq_field(self, modulus, limbs64):
        """random_long01Seq(FF): Limbs[2 N] -> BigInt[2 N 64] -> reduced mod the modulus (reduceViaMont computes exactly a mod p)"""
        return self.random_long01seq_bigint(2 * limbs64 * 64) % modulus
