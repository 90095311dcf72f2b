// pt_headless — headless frame-dump driver: the reference's frame loops (main.cu:284-303,
// main.cpp:153-249) without X11.  Builds a scene with the host mirror of the reference's
// interfaces, renders progressively through the C ABI, writes a P6 PPM (gamma 2.2, +0.5
// rounding — Camera::saveImage, cpu_pathtracer/include/Camera.h:162-180) and/or the raw
// float4 accumulation buffer (sum r,g,b, sample count) that pt_upload_accum can resume from.
#include "pt_host.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void usage() {
    std::puts(
        "usage: pt_headless [--scene cuda_default|cpu_default|glass_stress|random4096] [--width W] [--height H]\n"
        "                   [--spp N] [--spp-per-launch K] [--max-depth D] [--integrator path|direct] [--seed S]\n"
        "                   [--mis] [--light-colour] [--device I] [--resume raw] [--out image.ppm] [--raw accum.f32]\n"
        "                   [--frames-in-flight 1..8]  overlap consecutive launches (default: 2 for a render, 1 for --script; same image bit for bit)\n"
        "                   [--counters]  run the instrumented kernel: print the event counters and the path-length histogram\n"
        "                   [--script camera.txt]   lines: <frames> <mouse_dx> <mouse_dy> <keys: wasd u(p) c(down) or ->\n"
        "                                           (the CPU app's frame loop, main.cpp:153-237, 1 spp per frame;\n"
        "                                           --out then names a printf pattern, one PPM per script line)");
}

int main(int argc, char** argv) {
    std::string scene = "cuda_default", integrator, out, raw, resume, script;
    uint32_t W = 1280, H = 720, spp = 64, per = 0, depth = 0, seed = 0x5EED0001u, flags = 0, flight = 0;   // 0: not given
    int device = 0;
    bool counters = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        if (a == "--scene") scene = next();
        else if (a == "--width") W = (uint32_t)std::atoi(next());
        else if (a == "--height") H = (uint32_t)std::atoi(next());
        else if (a == "--spp") spp = (uint32_t)std::atoi(next());
        else if (a == "--spp-per-launch") per = (uint32_t)std::atoi(next());
        else if (a == "--max-depth") depth = (uint32_t)std::atoi(next());
        else if (a == "--integrator") integrator = next();
        else if (a == "--seed") seed = (uint32_t)std::strtoul(next(), nullptr, 0);
        else if (a == "--mis") flags |= PT_FLAG_NEE_MIS;
        else if (a == "--light-colour") flags |= PT_FLAG_LIGHT_COLOUR;
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--out") out = next();
        else if (a == "--raw") raw = next();
        else if (a == "--resume") resume = next();
        else if (a == "--script") script = next();
        else if (a == "--frames-in-flight") flight = (uint32_t)std::atoi(next());
        else if (a == "--counters") counters = true;
        else { usage(); return a == "--help" ? 0 : 2; }
    }
    try {
        PtSceneSoA s; PtCamera c;
        if (pt_builtin_scene(scene.c_str(), W, H, &s, &c) != PT_OK) { std::fprintf(stderr, "unknown scene %s\n", scene.c_str()); return 2; }
        if (integrator.empty()) integrator = scene == "cpu_default" ? "direct" : "path";
        PtRenderParams p = pt::defaultParams();
        p.integrator = integrator == "direct" ? PT_INTEGRATOR_DIRECT : PT_INTEGRATOR_PATH;
        p.seed = seed; p.flags = flags | (counters ? PT_FLAG_COUNTERS : 0u);
        p.max_depth = depth ? depth : (scene == "glass_stress" ? 32u : scene == "random4096" ? 16u : 50u);
        if (p.integrator == PT_INTEGRATOR_PATH) c.model = PT_CAMERA_CUDA;
        if (per == 0) per = spp < 16 ? spp : 16;

        pt::Renderer r(device);
        r.setScene(s);
        r.setCamera(c);
        // default: two frames in flight for a render (nothing reads the buffer between its launches), one for a scripted
        // session (every frame is shown: the library renders the next one ahead, pt_set_render_ahead)
        if (flight == 0) flight = script.empty() ? 2u : 1u;
        if (flight > 1) r.setFramesInFlight(flight);
        const size_t npix = (size_t)W * H;
        uint32_t begin = 0;
        if (!resume.empty()) {  // raw (sum, count) checkpoint
            std::vector<float> acc(npix * 4);
            FILE* f = std::fopen(resume.c_str(), "rb");
            if (!f || std::fread(acc.data(), 4, acc.size(), f) != acc.size()) { std::fprintf(stderr, "cannot read %s\n", resume.c_str()); return 1; }
            std::fclose(f);
            pt::check(pt_upload_accum(r.ctx(), acc.data(), acc.size() * 4));
            begin = (uint32_t)acc[3];
        }
        if (!script.empty()) {
            // The CPU app's frame loop (main.cpp:153-237) fed from a file instead of X11 events: one
            // sample per pixel per frame; any mouse motion or movement key re-aims the camera and
            // resets the accumulation (:196-205); frame_count counts frames since the last reset.
            FILE* f = std::fopen(script.c_str(), "r");
            if (!f) { std::fprintf(stderr, "cannot read %s\n", script.c_str()); return 1; }
            pt::FlyCamera fly(c);
            uint32_t frame_count = 0, total = 0, line = 0, resets = 0;
            unsigned frames; double dx, dy; char keys[32];
            r.sync();
            auto t0 = std::chrono::steady_clock::now();
            while (std::fscanf(f, "%u %lf %lf %31s", &frames, &dx, &dy, keys) == 4) {
                uint32_t mask = 0;
                for (const char* k = keys; *k; ++k)
                    mask |= *k == 'w' ? PT_KEY_W : *k == 's' ? PT_KEY_S : *k == 'a' ? PT_KEY_A : *k == 'd' ? PT_KEY_D
                          : *k == 'u' ? PT_KEY_SPACE : *k == 'c' ? PT_KEY_CTRL : 0u;
                for (unsigned i = 0; i < frames; ++i) {
                    if (fly.step(dx, dy, mask)) {
                        r.setCamera(fly.camera());
                        pt::check(pt_reset(r.ctx()));
                        frame_count = 0; ++resets;
                    } else if (total == 0) {
                        r.setCamera(fly.camera());  // the loop's first lookAt re-aims the camera even without input (:168)
                    }
                    r.render(frame_count, 1, p);
                    ++frame_count; ++total;
                }
                if (!out.empty()) {
                    char name[512];
                    std::snprintf(name, sizeof name, out.c_str(), line);
                    r.resolve(PT_RESOLVE_RGB8_PPM);
                    std::vector<unsigned char> img(npix * 3);
                    r.readback(img.data(), img.size());
                    FILE* o = std::fopen(name, "wb");
                    if (!o) { std::fprintf(stderr, "cannot write %s\n", name); return 1; }
                    std::fprintf(o, "P6\n%u %u\n255\n", W, H);
                    std::fwrite(img.data(), 1, img.size(), o);
                    std::fclose(o);
                }
                ++line;
            }
            std::fclose(f);
            r.sync();
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("{\"scene\": \"%s\", \"script_lines\": %u, \"frames\": %u, \"resets\": %u, \"frames_since_reset\": %u, "
                        "\"seconds\": %.6f, \"fps\": %.2f}\n", scene.c_str(), line, total, resets, frame_count, sec, total / sec);
            if (!raw.empty()) {
                std::vector<float> acc(npix * 4);
                r.readbackAccum(acc.data(), acc.size() * 4);
                FILE* o = std::fopen(raw.c_str(), "wb");
                if (!o) { std::fprintf(stderr, "cannot write %s\n", raw.c_str()); return 1; }
                std::fwrite(acc.data(), 4, acc.size(), o);
                std::fclose(o);
            }
            return 0;
        }
        r.sync();
        auto t0 = std::chrono::steady_clock::now();
        for (uint32_t sdone = 0; sdone < spp; sdone += per) r.render(begin + sdone, std::min(per, spp - sdone), p);
        r.sync();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("{\"scene\": \"%s\", \"integrator\": \"%s\", \"width\": %u, \"height\": %u, \"spp\": %u, \"seconds\": %.6f, "
                    "\"msamples_per_s\": %.3f}\n", scene.c_str(), integrator.c_str(), W, H, spp, sec, npix * (double)spp / sec / 1e6);
        if (counters) {   // the event counters that feed the FLOP model and the bounce histogram (§8(d).4)
            PtCounters k; uint64_t h[PT_HISTOGRAM_BINS];
            pt::check(pt_counters(r.ctx(), &k));
            pt::check(pt_path_length_histogram(r.ctx(), h));
            std::printf("{\"counters\": {\"samples\": %llu, \"rays\": %llu, \"hits\": %llu, \"sphere_roots\": %llu, \"bounces\": [%llu, %llu, %llu, %llu, %llu], "
                        "\"diffuse_updates\": %llu, \"nee_samples\": %llu, \"shadow_rays\": %llu}, \"path_length_histogram\": [",
                        (unsigned long long)k.samples, (unsigned long long)k.rays, (unsigned long long)k.hits, (unsigned long long)k.sphere_roots,
                        (unsigned long long)k.bounces[0], (unsigned long long)k.bounces[1], (unsigned long long)k.bounces[2], (unsigned long long)k.bounces[3],
                        (unsigned long long)k.bounces[4], (unsigned long long)k.diffuse_updates, (unsigned long long)k.nee_samples, (unsigned long long)k.shadow_rays);
            for (uint32_t b = 0; b < PT_HISTOGRAM_BINS; ++b) std::printf("%s%llu", b ? ", " : "", (unsigned long long)h[b]);
            std::printf("]}\n");
        }
        if (!out.empty()) {
            r.resolve(PT_RESOLVE_RGB8_PPM);
            std::vector<unsigned char> img(npix * 3);
            r.readback(img.data(), img.size());
            FILE* f = std::fopen(out.c_str(), "wb");
            if (!f) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
            std::fprintf(f, "P6\n%u %u\n255\n", W, H);
            std::fwrite(img.data(), 1, img.size(), f);
            std::fclose(f);
        }
        if (!raw.empty()) {
            std::vector<float> acc(npix * 4);
            r.readbackAccum(acc.data(), acc.size() * 4);
            FILE* f = std::fopen(raw.c_str(), "wb");
            if (!f) { std::fprintf(stderr, "cannot write %s\n", raw.c_str()); return 1; }
            std::fwrite(acc.data(), 4, acc.size(), f);
            std::fclose(f);
        }
    } catch (const pt::Error& e) {
        std::fprintf(stderr, "pt error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
