// host_internal.h — shared by the host pipeline translation units.
#pragma once

#include <cstdarg>
#include <string>
#include <vector>

#include "pthost.h"

namespace pth {

// Records a thread-local error message; always returns 1 so callers can
// `return set_error(...)`.
int set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

bool read_file(const std::string &path, std::vector<uint8_t> &out);

} // namespace pth

// Owning scene: every pointer in `desc` points into these vectors.
struct pth_scene {
    std::vector<pt_vertex> vertices;
    std::vector<pt_face> faces;
    std::vector<pt_light> lights;
    std::vector<pt_material> materials;
    std::vector<pt_texture> textures;
    std::vector<std::vector<uint8_t>> texel_storage; // one blob per texture
    pt_scene_desc desc{};
    void refresh_desc();
};
