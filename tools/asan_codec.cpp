// Host-side sanitizer harness for the image decoders (untrusted input): every file given is decoded from an exact-size heap
// copy under AddressSanitizer + UBSan.  Built and run by tests/test_host_surface.py::test_decoders_under_address_sanitizer.
#include "../rust-wgpu-raytracing_amd/host/image_codec.hpp"
#include <cstdio>
#include <fstream>
#include <iterator>
int main(int argc, char **argv) {
    for (int i = 1; i < argc; i++) {
        std::ifstream f(argv[i], std::ios::binary);
        std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        // exact-size heap copy so that ASan sees an overread of even one byte
        uint8_t *p = new uint8_t[d.size()];
        std::copy(d.begin(), d.end(), p);
        rwr::codec::Image img; std::string err;
        bool ok = rwr::codec::decode_image(p, d.size(), img, err);
        std::printf("%s: %s %s\n", argv[i], ok ? "ok" : "rejected:", err.c_str());
        delete[] p;
    }
}
