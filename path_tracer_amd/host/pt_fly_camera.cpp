// pt_fly_camera.cpp -- the CPU app's per-frame camera update (cpu_pathtracer/src/main.cpp:157-195)
// as a pure host function of the C ABI.  Compiled with -ffp-contract=off: every operation is one
// IEEE double operation in the order the reference writes them.
#include "../../include/pt.h"
#include "../csrc/pt_error.h"

#include <algorithm>
#include <cmath>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

extern "C" int pt_fly_camera_step(PtCamera* cam, double* yaw, double* pitch, double mouse_dx, double mouse_dy, uint32_t keys) {
    pt_clear_error();
    if (!cam || !yaw || !pitch) return pt_fail(PT_ERR_INVALID, "pt_fly_camera_step: camera, yaw and pitch must not be null");
    const double mouse_sensitivity = 0.002, move_speed = 0.1;  // main.cpp:137-138
    *yaw += mouse_dx * mouse_sensitivity;                                                                    // :157
    *pitch = std::clamp(*pitch - mouse_dy * mouse_sensitivity, -M_PI / 2.0 + 0.1, M_PI / 2.0 - 0.1);        // :158-159
    const double f[3] = {std::cos(*pitch) * std::cos(*yaw), std::sin(*pitch), std::cos(*pitch) * std::sin(*yaw)};  // :162-166
    // right = cross(forward, (0,1,0)).normalized()  (:171)
    double r[3] = {f[1] * 0.0 - f[2] * 1.0, f[2] * 0.0 - f[0] * 0.0, f[0] * 1.0 - f[1] * 0.0};
    const double len = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (len > 0) { const double inv = 1.0 / len; r[0] *= inv; r[1] *= inv; r[2] *= inv; }  // Vec3.h:48-63
    double pos[3] = {cam->origin[0], cam->origin[1], cam->origin[2]};
    auto add = [&](const double* v, double s) { for (int k = 0; k < 3; ++k) pos[k] = pos[k] + v[k] * s; };
    const double up[3] = {0, 1, 0};
    if (keys & PT_KEY_W) add(f, move_speed);       // :174-191
    if (keys & PT_KEY_S) add(f, -move_speed);
    if (keys & PT_KEY_A) add(r, -move_speed);
    if (keys & PT_KEY_D) add(r, move_speed);
    if (keys & PT_KEY_SPACE) add(up, move_speed);
    if (keys & PT_KEY_CTRL) add(up, -move_speed);
    for (int k = 0; k < 3; ++k) { cam->origin[k] = pos[k]; cam->target[k] = pos[k] + f[k]; cam->up[k] = up[k]; }  // :193
    cam->normalize_up = 0;
    const uint32_t move_keys = PT_KEY_W | PT_KEY_S | PT_KEY_A | PT_KEY_D | PT_KEY_SPACE | PT_KEY_CTRL;
    return (mouse_dx != 0 || mouse_dy != 0 || (keys & move_keys) != 0) ? 1 : 0;                               // :196-199
}
