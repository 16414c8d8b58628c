// java_random.h -- host-side java.util.Random, so that the device weights start out exactly as
// the reference's appendLayer leaves them (SCE:111, SCE:147-151: `new Random(1)`,
// `nextDouble() - 0.5` per weight, layer by layer, row-major).  The generator is the 48-bit
// LCG specified by the JDK Javadoc of java.util.Random; only next(bits), nextDouble and
// nextInt(bound) are needed on this path (NNT:152 draws batch indices with nextInt(bound)).
#pragma once
#include <cstdint>

namespace gnn {

class JavaRandom {
public:
    explicit JavaRandom(int64_t seed) { set_seed(seed); }
    void set_seed(int64_t seed) { state_ = (static_cast<uint64_t>(seed) ^ kMul) & kMask; }

    int32_t next(int bits) {
        state_ = (state_ * kMul + kAdd) & kMask;
        return static_cast<int32_t>(static_cast<uint32_t>(state_ >> (48 - bits)));
    }
    double next_double() {
        const int64_t hi = next(26), lo = next(27);
        return static_cast<double>((hi << 27) + lo) * 0x1.0p-53;
    }
    int32_t next_int(int32_t bound) {
        int32_t r = next(31);
        const int32_t m = bound - 1;
        if ((bound & m) == 0) return static_cast<int32_t>((static_cast<int64_t>(bound) * r) >> 31);
        for (int32_t u = r;; u = next(31)) {
            r = u % bound;
            // Java: u - r + m < 0 with int wrap-around
            if (static_cast<int32_t>(static_cast<uint32_t>(u) - static_cast<uint32_t>(r) +
                                     static_cast<uint32_t>(m)) >= 0)
                return r;
        }
    }

private:
    static constexpr uint64_t kMul = 0x5DEECE66DULL, kAdd = 0xBULL, kMask = (1ULL << 48) - 1;
    uint64_t state_;
};

} // namespace gnn
