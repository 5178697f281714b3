// Randomised check of the PLY loader's fast decimal -> f32 path against strtof (both must be correctly rounded, like ply-rs / Rust).
// Built and run by tests/test_host.py; includes the translation unit to reach the file-local function.
#include "../../raytracer_amd/csrc/host/scene.cpp"
#include <random>
int main() {
    std::mt19937_64 g(7); char buf[64]; long bad = 0, n = 0;
    for (int it = 0; it < 3000000; ++it) {
        int mode = it % 5; 
        if (mode == 0) { double v = std::ldexp((double)(g() >> 11), (int)(g() % 80) - 100); snprintf(buf, sizeof buf, "%.*g", 1 + (int)(g() % 17), (g() & 1) ? -v : v); }
        else if (mode == 1) { float f; uint32_t b = (uint32_t)g(); memcpy(&f, &b, 4); if (!(f - f == 0)) continue; snprintf(buf, sizeof buf, "%.*g", 1 + (int)(g() % 12), (double)f); }
        else if (mode == 2) { snprintf(buf, sizeof buf, "%.*f", (int)(g() % 10), ((double)(int64_t)(g() % 2000000) - 1000000) / 7919.0); }
        else if (mode == 3) { /* exact f32 midpoints */ uint32_t b = (uint32_t)(g() % 0x7f000000u); float f0; memcpy(&f0, &b, 4); ++b; float f1; memcpy(&f1, &b, 4); double m = ((double)f0 + (double)f1) / 2; snprintf(buf, sizeof buf, "%.17g", m); }
        else { snprintf(buf, sizeof buf, "%lld.%llue%d", (long long)(g() % 1000), (unsigned long long)(g() % 100000000), (int)(g() % 60) - 30); }
        float a = rth::parse_f32(buf), b = strtof(buf, nullptr); ++n;
        if (memcmp(&a, &b, 4)) { if (bad++ < 10) printf("MISMATCH %s -> %a vs %a\n", buf, a, b); }
    }
    printf("%ld tested, %ld mismatches\n", n, bad); return bad != 0;
}
