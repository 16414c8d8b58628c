// convert_helper_check.cpp -- the host batch's fp64 -> f32 conversion shared with helper threads (csrc/convert_helper.h),
// stressed on the CPU: sizes around the sharing threshold, back-to-back calls (helpers polling), pauses longer than the
// polling window (helpers asleep), and two callers at once (the second converts alone).  Built with -fsanitize=thread by
// tests/test_convert_helper.py; prints "ok <calls>" or the first mismatch.
#include "convert_helper.h"

#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

using gnn::host::ConvertHelper;

static bool one(ConvertHelper &c, std::mt19937_64 &rng, size_t n) {
    std::vector<double> src(n);
    std::vector<float> dst(n + 16, -7.f);
    std::uniform_real_distribution<double> u(-3.0, 3.0);
    for (auto &x : src) x = u(rng);
    c.run(src.data(), dst.data(), n);
    for (size_t i = 0; i < n; i++)
        if (dst[i] != (float)src[i]) { printf("mismatch n=%zu i=%zu\n", n, i); return false; }
    for (size_t i = n; i < n + 16; i++)
        if (dst[i] != -7.f) { printf("overrun n=%zu i=%zu\n", n, i); return false; }
    return true;
}

int main() {
    ConvertHelper c;
    std::mt19937_64 rng(12345);
    const size_t sizes[] = {1, 100, 32767, 32768, 32769, 100352, 100353, 65551, 401408};
    int calls = 0;
    for (int round = 0; round < 40; round++) {
        for (size_t n : sizes) {
            if (!one(c, rng, n)) return 1;
            calls++;
        }
        if (round % 8 == 7) std::this_thread::sleep_for(std::chrono::milliseconds(5)); // the helpers go to sleep
    }
    // two callers at once: whoever finds the pool busy converts alone
    bool ok2 = true;
    std::thread other([&] {
        std::mt19937_64 r2(777);
        for (int i = 0; i < 100 && ok2; i++) ok2 = one(c, r2, 100352);
    });
    for (int i = 0; i < 100; i++) {
        if (!one(c, rng, 100352)) return 1;
        calls++;
    }
    other.join();
    if (!ok2) return 1;
    printf("ok %d\n", calls + 100);
    return 0;
}
