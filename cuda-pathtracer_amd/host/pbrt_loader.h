// pbrt_loader.h — PBRT (v3 syntax) scene import with the reference importer's rules
// (include/utils/pbrt_loader.h:178-422 over the vendored pbrtParser, see pbrt_loader.cpp).
#pragma once
#include <string>
#include <vector>

#include "primitive.h"

namespace ptmi {

// loadPBRT — utils/pbrt_loader.h:178-422: triangle meshes of the flattened scene -> triangles (vertices through the shape's
// and the instance's transforms, first-vertex normal or geometric normal, material -> single albedo, diffuse RGB area light
// -> Le); more than 2,000,000 triangles -> a 12-triangle bounding-box proxy.  Returns false (message in *error) where the
// reference returns false or where it throws / crashes.
bool loadPBRT(const std::string& pbrt_filename, std::vector<Primitive>& out, std::string* error = nullptr);

}  // namespace ptmi
