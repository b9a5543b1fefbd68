// rwr_render — headless counterpart of the reference's `run()` event loop
// (/root/reference/src/lib.rs:1233-1352, src/main.rs): State::new, then per frame
// input* -> update -> render, with key events taken from a script instead of a window.
//
//   rwr_render --res DIR [--scene suzanne_lowpoly.obj] [--size 600x600] [--keys "S*15,D*4"]
//              [--frames N] [--resize WxH@FRAME]... [--spp N] [--bounces B] [--out frame.png] [--time]
//
// --keys: comma separated KEY*COUNT; each entry holds KEY down for COUNT frames
// (KEY in W A S D Up Down Left Right Space LShift, or '-' for no key).  After the script,
// --frames more frames are rendered with no key held.  The window default is 600x600
// (lib.rs:1248-1251).  --resize WxH@FRAME (repeatable): a WindowEvent::Resized delivered before frame FRAME
// (0-based, counted over the whole run) -> State::resize (lib.rs:772-989, 1323-1330), including its quirk: the
// camera's aspect is recomputed from the size BEFORE the resize.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "state.hpp"

using namespace rwr;

static VirtualKeyCode parse_key(const std::string &k)
{
    if (k == "W") return VirtualKeyCode::W;
    if (k == "A") return VirtualKeyCode::A;
    if (k == "S") return VirtualKeyCode::S;
    if (k == "D") return VirtualKeyCode::D;
    if (k == "Up") return VirtualKeyCode::Up;
    if (k == "Down") return VirtualKeyCode::Down;
    if (k == "Left") return VirtualKeyCode::Left;
    if (k == "Right") return VirtualKeyCode::Right;
    if (k == "Space") return VirtualKeyCode::Space;
    if (k == "LShift") return VirtualKeyCode::LShift;
    return VirtualKeyCode::Other;
}

int main(int argc, char **argv)
{
    std::string res, scene = "suzanne_lowpoly.obj", out, keys;
    uint32_t w = 600, h = 600, frames = 1, spp = 1, bounces = 0;
    bool timing = false;
    struct Resize { uint64_t frame; uint32_t w, h; };
    std::vector<Resize> resizes;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        if (a == "--res") res = next();
        else if (a == "--scene") scene = next();
        else if (a == "--out") out = next();
        else if (a == "--keys") keys = next();
        else if (a == "--frames") frames = (uint32_t)std::atoi(next());
        else if (a == "--spp") spp = (uint32_t)std::atoi(next());
        else if (a == "--bounces") bounces = (uint32_t)std::atoi(next());
        else if (a == "--time") timing = true;
        else if (a == "--resize") {
            Resize r{0, 0, 0};
            unsigned long long f = 0;
            if (std::sscanf(next(), "%ux%u@%llu", &r.w, &r.h, &f) != 3) { std::fprintf(stderr, "--resize WxH@FRAME\n"); return 2; }
            r.frame = f;
            resizes.push_back(r);
        }
        else if (a == "--size") {
            if (std::sscanf(next(), "%ux%u", &w, &h) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; }
        } else if (a == "--help" || a == "-h") {
            std::printf("usage: rwr_render --res DIR [--scene F.obj] [--size WxH] [--keys \"S*15,D*4\"] [--frames N] "
                        "[--resize WxH@FRAME]... [--spp N] [--bounces B] [--out frame.png] [--time]\n");
            return 0;
        } else {
            std::fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (res.empty()) { std::fprintf(stderr, "--res DIR is required (the reference bakes OUT_DIR/res in at compile time)\n"); return 2; }

    // script: (key, frames held)
    std::vector<std::pair<VirtualKeyCode, uint32_t>> script;
    size_t pos = 0;
    while (pos < keys.size()) {
        const size_t comma = keys.find(',', pos);
        const std::string item = keys.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        const size_t star = item.find('*');
        const std::string k = item.substr(0, star);
        const uint32_t n = star == std::string::npos ? 1u : (uint32_t)std::atoi(item.c_str() + star + 1);
        if (k != "-" && parse_key(k) == VirtualKeyCode::Other) { std::fprintf(stderr, "unknown key '%s'\n", k.c_str()); return 2; }
        script.emplace_back(parse_key(k), n);
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }

    try {
        State state(w, h, res, scene);
        const rwr_render_params params{spp, bounces, 0u, 0u};
        uint64_t rendered = 0;
        const auto t0 = std::chrono::steady_clock::now();
        auto frame = [&]() {  // [Resized: resize()] then RedrawRequested: update() then render() (lib.rs:1323-1337)
            for (const Resize &r : resizes)
                if (r.frame == rendered) state.resize(r.w, r.h);
            state.update();
            state.render(&params);
            rendered++;
        };
        for (const auto &[key, n] : script) {
            if (key != VirtualKeyCode::Other) state.input(KeyboardInput{ElementState::Pressed, key});
            for (uint32_t f = 0; f < n; f++) frame();
            if (key != VirtualKeyCode::Other) state.input(KeyboardInput{ElementState::Released, key});
        }
        for (uint32_t f = 0; f < frames; f++) frame();
        check(rwr_synchronize(state.context()));
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const Camera &c = state.camera();
        std::printf("frames %llu  eye (%.6f, %.6f, %.6f)  target (%.6f, %.6f, %.6f)  size %ux%u  aspect %.9g\n", (unsigned long long)rendered,
                    c.eye.x, c.eye.y, c.eye.z, c.target.x, c.target.y, c.target.z, state.size().width, state.size().height, (double)c.aspect);
        if (timing) std::printf("%.3f ms/frame over %llu frames (update + render, host wall clock)\n", sec * 1e3 / (double)rendered,
                                (unsigned long long)rendered);
        if (!out.empty()) {
            state.present(out);
            std::printf("wrote %s (%ux%u, row 0 of the framebuffer at the bottom, sRGB encoded)\n", out.c_str(), state.size().width,
                        state.size().height);
        }
    } catch (const RwrFailure &e) {
        std::fprintf(stderr, "rwr_render: error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
