/*
 * rtc_host.h — C entry points of librtc_host.so: the steps either side of the
 * hot path (scene JSON / OBJ / PNG loading and flattening before it, Canvas
 * output after it), for callers that are not C++.  The per-pixel work is
 * reached only through include/rtc.h; nothing here renders on the CPU.
 *
 * Reference counterparts:
 *   rtch_scene_load      parseScene, src/parsing/scene.zig:612-661 (+ obj.zig, zigimg PNG decode), then the
 *                        depth-first flattening into rtc_scene_desc
 *   rtch_scene_camera    SceneInfo.camera (scene.zig:633-648); width/height override the file's values
 *   rtch_camera_rotate   Renderer.rotateCamera, src/lib.zig:166-178
 *   rtch_camera_move     Renderer.moveCamera,   src/lib.zig:180-190
 *   rtch_camera_make     Camera.new + Matrix.viewTransform, camera.zig:33-61, matrix.zig:54-67
 *   rtch_canvas_ppm      Canvas.ppm, canvas.zig:181-254
 *   rtch_canvas_rgba8    the RGBA8 framebuffer of lib.zig:146-153 (clamp, color.zig:61-71)
 *   rtch_scene_render    main.zig:92: load -> Camera.render -> Canvas, through rtc_scene_create / rtc_render
 *
 * Every function that returns int returns 0 on success; otherwise rtch_last_error()
 * holds "<ZigStyleErrorName>: detail" (thread-local).
 */
#ifndef RTC_HOST_H
#define RTC_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "rtc.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *rtch_last_error(void);

/* Parses a scene description; files it names (OBJ meshes, PNG textures) are read from data_dir + name. */
int rtch_scene_load(const char *scene_json, const char *data_dir, void **out_handle);
/* Threads rtch_scene_load may build the scene's objects on: 0 (the default) = what the process may use, at most 16;
 * 1 = one loop over "objects", as scene.zig:650-655.  The description is the same to the bit either way. */
void rtch_set_loader_threads(uint32_t threads);
void rtch_scene_free(void *handle);
/* The flattened World, valid while the handle lives; pass it to rtc_scene_create. */
const rtc_scene_desc *rtch_scene_desc(void *handle);
int rtch_scene_camera(void *handle, uint32_t width, uint32_t height, rtc_camera *out);
int rtch_camera_rotate(void *handle, double angle);
int rtch_camera_move(void *handle, double distance);
int rtch_camera_make(uint32_t hsize, uint32_t vsize, double fov, const double from[3], const double to[3],
                     const double up[3], rtc_camera *out);
/* Returns the number of bytes the PPM needs; writes at most cap bytes. */
size_t rtch_canvas_ppm(const double *rgb, uint32_t w, uint32_t h, char *buf, size_t cap);
void rtch_canvas_rgba8(const double *rgb, uint32_t w, uint32_t h, uint8_t *out);
int rtch_scene_render(void *handle, uint32_t width, uint32_t height, uint32_t max_depth, double *rgb_out);

#ifdef __cplusplus
}
#endif
#endif /* RTC_HOST_H */
