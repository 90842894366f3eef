// sensor.h — pin-hole camera, host side (include/rendering/sensor.h:14-100).
//
// The device only ever needs origin / lower_left_corner / horizontal / vertical
// (get_ray, sensor.h:31-33); they are derived here with the reference's float/double
// mix and shipped to the kernels by value.  Trigonometry goes through the shared
// numerics contract (include/ptmi_math.h) instead of the host libm so that the frame
// is identical on every machine.
#pragma once
#include "../../include/ptmi_math.h"
#include "../csrc/pt_vec.h"

namespace ptmi {

struct CameraFrame { f3 origin, lower_left_corner, horizontal, vertical; };

class Sensor {
public:
    Sensor() : Sensor(mk3(0.5f, 3.0f, 8.5f), mk3(0.0f, 2.5f, 0.0f), mk3(0.0f, 1.0f, 0.0f), 40.0f, 1.0f) {}
    Sensor(f3 lookfrom, f3 lookat_, f3 vup_, float vfov_, float aspect_) {       // sensor.h:16-29
        origin = lookfrom; vup = vup_; lookat = lookat_; vfov = vfov_; aspect = aspect_;
        radius = length(lookfrom - lookat_);
        yaw = 90.0f; pitch = 0.0f;
        updateCamera();
    }

    void updateCamera() {                                                        // sensor.h:38-51
        const float theta = (float)((double)vfov * PTMI_PI_D / (double)180.0f);  // M_PI is a double there
        const float half_height = (float)ptmi_tan_d((double)(theta / 2.0f));
        const float half_width = aspect * half_height;
        const f3 w = unit_vector(origin - lookat);
        const f3 u = unit_vector(cross(vup, w));
        const f3 v = cross(w, u);
        lower_left_corner = origin - half_width * u - half_height * v - w;
        horizontal = (2 * half_width) * u;
        vertical = (2 * half_height) * v;
    }

    void updateCameraOrbit() {                                                   // sensor.h:56-67
        const float yawRad = toRadian(yaw), pitchRad = toRadian(pitch);
        float sy, cy, sp, cp;
        ptmi_sincosf(yawRad, &sy, &cy);
        ptmi_sincosf(pitchRad, &sp, &cp);
        origin.x = lookat.x + radius * cp * cy;
        origin.y = lookat.y + radius * sp;
        origin.z = lookat.z + radius * cp * sy;
        updateCamera();
    }

    CameraFrame frame() const { return CameraFrame{origin, lower_left_corner, horizontal, vertical}; }

    float radius, yaw, pitch;
    f3 vup, lookat, origin, lower_left_corner, horizontal, vertical;
    float vfov, aspect;
    int image_width = 0, image_height = 0;

private:
    static float toRadian(float deg) { return (float)((double)deg * PTMI_PI_D / (double)180.0f); }   // sensor.h:11
};

}  // namespace ptmi
