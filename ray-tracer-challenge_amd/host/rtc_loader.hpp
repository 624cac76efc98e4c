// rtc_loader.hpp — scene JSON + OBJ loaders (the step immediately before the hot
// path).  Restates src/parsing/scene.zig and
// src/parsing/obj.zig; see rtc_loader.cpp for the line-by-line citations.
#pragma once
#include <functional>
#include <map>
#include <optional>
#include <string>
#include <string_view>
#include <vector>

#include "rtc_scene.hpp"

namespace rtc {

// scene.zig:612-618 `load_file_data` callback: file name -> bytes.
using FileLoader = std::function<std::string(const std::string& file_name)>;
FileLoader directoryLoader(const std::string& dir);  // main.zig:14-21 ("data/" + name)

struct SceneInfo {  // scene.zig:608-610
  Camera camera;
  World world;
};

// scene.zig:612-661.  Throws rtc::Error whose .name is the Zig error name
// (UnknownDefinition, NotInvertible, MissingField, UnknownField, ...).
SceneInfo parseScene(const std::string& scene_json, const FileLoader& load_file_data);
// The entries of "objects" are independent of each other (definitions are read-only, an object's transform, material and
// divide(8) are its own): parseScene builds them on up to this many threads - 0: what the process may use (CPU affinity,
// cgroup quota), at most 16; 1: the reference's one loop.  The World is the same to the bit whatever the count (ids
// included: ShapeIdScope); `load_file_data` must then be callable from several threads at once (directoryLoader is).
void setLoaderThreads(unsigned threads);
unsigned loaderThreads();

// obj.zig:11-286
class ObjParser {
 public:
  struct InheritedState {  // obj.zig:186-189
    std::optional<Material> material;
    std::optional<bool> casts_shadow;
  };

  ObjParser();
  void loadObj(const std::string& obj, const InheritedState& state, bool normalize);  // obj.zig:191-279
  Shape toGroup() const { return default_group; }                                      // obj.zig:281-283
  Shape takeGroup() { return std::move(default_group); }  // the same, for a parser that is done (no copy of a kilobyte per triangle)

  Shape default_group;
  std::map<std::string, size_t> named_groups;  // name -> index in default_group.children
  Tuple offset = Tuple::vec3(0.0, 0.0, 0.0);
  double scale = 1.0;
  std::vector<Tuple> vertices;
  std::vector<Tuple> normals;
  size_t lines_ignored = 0;

 private:
  long active_group_ = -1;  // -1: default group, else index into default_group.children
  Shape& activeGroup() { return active_group_ < 0 ? default_group : default_group.children[active_group_]; }
  void handleLine(std::string_view line, const InheritedState& state);
  std::vector<std::string_view> line_tokens_;  // (scratch of handleLine)
};

}  // namespace rtc
