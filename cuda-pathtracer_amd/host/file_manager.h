// file_manager.h — OBJ/MTL scene loader with the reference loader's rules
// (include/utils/file_manager.h:39-273).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "primitive.h"

namespace ptmi {

struct Material {                       // file_manager.h:27-30
    f3 bsdf = {0.8f, 0.8f, 0.8f};
    f3 Le = {0.0f, 0.0f, 0.0f};
};

// loadMTL — file_manager.h:39-79.  Only Kd and Ke are read; a missing file yields an empty table.
std::map<std::string, Material> loadMTL(const std::string& filename);

// loadOBJ — file_manager.h:93-273.  Returns false if the file cannot be opened or holds no valid face.
bool loadOBJ(const std::string& obj_filename, std::vector<Primitive>& out, std::string* warnings = nullptr);

// convertQuadsToTriangles — application_state.h:323-365
std::vector<Primitive> convertQuadsToTriangles(const std::vector<Primitive>& primitives);

// subdivide_primitives — rendering/form_factors.h:520-574
std::vector<Primitive> subdivide_primitives(const std::vector<Primitive>& prims, int num_subdivisions);

// "Save PNG" (ui/ui_windows.h:195-210: stbi_flip_vertically_on_write(1); stbi_write_png(path, w, h, 3, h_image, w * 3)):
// 8-bit RGB, rows flipped so that the frame's bottom row (row 0 of the image buffers) ends up last in the file.
// Own writer (zlib stream of stored blocks); the vendored stb_image_write is third-party code and not reproduced.
bool writePNG(const std::string& path, int width, int height, const unsigned char* rgb8_bottom_up);

}  // namespace ptmi
