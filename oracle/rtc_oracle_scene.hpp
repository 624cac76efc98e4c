// rtc_oracle_scene.hpp — CPU ORACLE, scene construction.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// An independent restatement of how the reference turns a scene JSON (+ OBJ files) into its Camera and World, so that
// the oracle no longer has to be fed the PRODUCT loader's flat description: with this header the oracle builds its own
// Shape tree from the same bytes the product loader reads, and tests/test_oracle_scene_cpu.py compares the two results
// table by table, bit for bit (leaf order, ids, inverses, group boxes, child lists, triangles, materials, patterns,
// lights, camera).  It shares no code with ray-tracer-challenge_amd/host/: its own JSON reader, the oracle's own Matrix.
//
// Follows, statement by statement (all file:line citations are into /root/reference/src/):
//   parsing/scene.zig    :164-190 inherit, :214-241 parseTransform, :243-405 parseUvPattern / parsePattern,
//                        :407-430 parseMaterial, :440-591 parseObject, :593-606 parseLight, :612-661 parseScene
//   parsing/obj.zig      :53-283 (handleVertex / VertexNormal / Face / NamedGroup, loadObj, toGroup)
//   raytracer/shapes/shape.zig  :123-130 the id counter, :235-284 group / csg constructors, :286-310 setTransform,
//                        :353-369 bounds / parentSpaceBounds, :372-399 divide
//   raytracer/shapes/group.zig  :75-78 addChild, :85-135 partitionChildren / makeSubgroup
//   raytracer/shapes/bounding_box.zig :24-110 add / contains / merge / transform / split
//   raytracer/shapes/{sphere,plane,cube,cylinder,cone,triangle,csg}.zig  bounds()
//   raytracer/camera.zig :33-61
// Every `Shape(T).new` of the reference bumps one process-wide id counter (shape.zig:123-130) - bounding boxes are
// Shapes too and take ids - and this file bumps orc::nextId() at exactly the same places, so the ids of the leaves come
// out as the reference's (relative to the counter's value when the scene's parse starts).
#pragma once
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <string>

#include "rtc_oracle.hpp"

namespace orc {
namespace scene {

// ------------------------------------------------------------------ a JSON reader of its own (std.json's data model)
struct Json {
  enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;  // in file order
  const Json* get(const char* key) const {
    for (const auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  const Json& at(const char* key) const {
    const Json* j = get(key);
    if (!j) throw std::runtime_error(std::string("MissingField: ") + key);
    return *j;
  }
};

class JsonReader {
 public:
  explicit JsonReader(const std::string& text) : s_(text) {}
  Json parse() {
    Json j = value();
    ws();
    if (i_ != s_.size()) throw std::runtime_error("SyntaxError: trailing characters");
    return j;
  }

 private:
  const std::string& s_;
  size_t i_ = 0;
  void ws() {
    while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_;
  }
  char peek() {
    ws();
    if (i_ >= s_.size()) throw std::runtime_error("UnexpectedEndOfInput");
    return s_[i_];
  }
  void expect(char c) {
    if (peek() != c) throw std::runtime_error(std::string("SyntaxError: expected ") + c);
    ++i_;
  }
  Json value() {
    const char c = peek();
    Json j;
    if (c == '{') {
      j.type = Json::Obj;
      ++i_;
      if (peek() == '}') {
        ++i_;
        return j;
      }
      for (;;) {
        Json k = string();
        expect(':');
        j.obj.emplace_back(k.str, value());
        if (peek() == ',') {
          ++i_;
          continue;
        }
        expect('}');
        return j;
      }
    }
    if (c == '[') {
      j.type = Json::Arr;
      ++i_;
      if (peek() == ']') {
        ++i_;
        return j;
      }
      for (;;) {
        j.arr.push_back(value());
        if (peek() == ',') {
          ++i_;
          continue;
        }
        expect(']');
        return j;
      }
    }
    if (c == '"') return string();
    if (s_.compare(i_, 4, "true") == 0) {
      i_ += 4;
      j.type = Json::Bool;
      j.b = true;
      return j;
    }
    if (s_.compare(i_, 5, "false") == 0) {
      i_ += 5;
      j.type = Json::Bool;
      return j;
    }
    if (s_.compare(i_, 4, "null") == 0) {
      i_ += 4;
      return j;
    }
    // a number: std.json hands the token to std.fmt.parseFloat, which rounds correctly, as strtod does
    const char* begin = s_.c_str() + i_;
    char* end = nullptr;
    j.num = std::strtod(begin, &end);
    if (end == begin) throw std::runtime_error("SyntaxError: value");
    i_ += static_cast<size_t>(end - begin);
    j.type = Json::Num;
    return j;
  }
  Json string() {
    expect('"');
    Json j;
    j.type = Json::Str;
    while (i_ < s_.size() && s_[i_] != '"') {
      if (s_[i_] == '\\' && i_ + 1 < s_.size()) {
        const char e = s_[i_ + 1];
        i_ += 2;
        switch (e) {
          case 'n': j.str += '\n'; break;
          case 't': j.str += '\t'; break;
          case 'r': j.str += '\r'; break;
          case 'b': j.str += '\b'; break;
          case 'f': j.str += '\f'; break;
          case 'u': {  // (scene files are ASCII; a BMP code point as UTF-8)
            const unsigned cp = static_cast<unsigned>(std::strtoul(s_.substr(i_, 4).c_str(), nullptr, 16));
            i_ += 4;
            if (cp < 0x80) {
              j.str += static_cast<char>(cp);
            } else if (cp < 0x800) {
              j.str += static_cast<char>(0xC0 | (cp >> 6));
              j.str += static_cast<char>(0x80 | (cp & 0x3F));
            } else {
              j.str += static_cast<char>(0xE0 | (cp >> 12));
              j.str += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
              j.str += static_cast<char>(0x80 | (cp & 0x3F));
            }
            break;
          }
          default: j.str += e;
        }
      } else {
        j.str += s_[i_++];
      }
    }
    if (i_ >= s_.size()) throw std::runtime_error("UnexpectedEndOfInput");
    ++i_;
    return j;
  }
};

// ------------------------------------------------------------------ bounding_box.zig:21-110 (build-time half)
// A BoundingBox is the payload of a Shape in the reference; `Shape(T).boundingBox()` goes through Shape.new and takes
// an id (shape.zig:123-130, :172-174): newBox() is that constructor.
struct Box {
  Tuple min = point(INF, INF, INF), max = point(-INF, -INF, -INF);
};
inline Box newBox() {
  (void)nextId();
  return Box{};
}
inline void boxAdd(Box& b, Tuple p) {  // :24-32; @min / @max return the non-NaN operand
  b.min.x = std::fmin(b.min.x, p.x);
  b.min.y = std::fmin(b.min.y, p.y);
  b.min.z = std::fmin(b.min.z, p.z);
  b.max.x = std::fmax(b.max.x, p.x);
  b.max.y = std::fmax(b.max.y, p.y);
  b.max.z = std::fmax(b.max.z, p.z);
}
inline bool boxContainsPoint(const Box& b, Tuple p) {  // :34-38
  return b.min.x <= p.x && p.x <= b.max.x && b.min.y <= p.y && p.y <= b.max.y && b.min.z <= p.z && p.z <= b.max.z;
}
inline bool boxContainsBox(const Box& b, const Box& o) { return boxContainsPoint(b, o.min) && boxContainsPoint(b, o.max); }  // :40-42
inline void boxMerge(Box& b, const Box& o) {  // :44-47
  boxAdd(b, o.min);
  boxAdd(b, o.max);
}
inline Box boxTransform(const Box& b, const Matrix& m) {  // :49-70
  const Tuple p1 = b.min;
  const Tuple p2 = point(b.min.x, b.min.y, b.max.z);
  const Tuple p3 = point(b.min.x, b.max.y, b.min.z);
  const Tuple p4 = point(b.min.x, b.max.y, b.max.z);
  const Tuple p5 = point(b.max.x, b.min.y, b.min.z);
  const Tuple p6 = point(b.max.x, b.min.y, b.max.z);
  const Tuple p7 = point(b.max.x, b.max.y, b.min.z);
  const Tuple p8 = b.max;
  Box n = newBox();
  boxAdd(n, m.tupleMul(p1));
  boxAdd(n, m.tupleMul(p2));
  boxAdd(n, m.tupleMul(p3));
  boxAdd(n, m.tupleMul(p4));
  boxAdd(n, m.tupleMul(p5));
  boxAdd(n, m.tupleMul(p6));
  boxAdd(n, m.tupleMul(p7));
  boxAdd(n, m.tupleMul(p8));
  return n;
}
inline void boxSplit(const Box& b, Box& left, Box& right) {  // :72-110
  const double dx = b.max.x - b.min.x;
  const double dy = b.max.y - b.min.y;
  const double dz = b.max.z - b.min.z;
  const double greatest = std::fmax(dx, std::fmax(dy, dz));
  double x0 = b.min.x, y0 = b.min.y, z0 = b.min.z;
  double x1 = b.max.x, y1 = b.max.y, z1 = b.max.z;
  if (greatest == dx) {
    x0 = x0 + dx / 2.0;
    x1 = x0;
  } else if (greatest == dy) {
    y0 = y0 + dy / 2.0;
    y1 = y0;
  } else {
    z0 = z0 + dz / 2.0;
    z1 = z0;
  }
  const Tuple mid_min = point(x0, y0, z0);
  const Tuple mid_max = point(x1, y1, z1);
  left = newBox();
  left.min = b.min;
  left.max = mid_max;
  right = newBox();
  right.min = mid_min;
  right.max = b.max;
}

// The tree under construction: orc::Shape plus what only the build needs (triangle corners for bounds()).  A group's
// _bbox is Shape::bmin / bmax.
inline Box groupBox(const Shape& s) {
  Box b;
  b.min = s.bmin;
  b.max = s.bmax;
  return b;
}
inline void setGroupBox(Shape& s, const Box& b) {
  s.bmin = b.min;
  s.bmax = b.max;
}

// Shape.bounds (shape.zig:353-362) -> the variant's bounds(): every leaf kind makes a fresh bounding-box Shape (an id);
// a group or csg returns a copy of its _bbox (no id).
inline Box shapeBounds(const Shape& s) {
  switch (s.kind) {
    case SPHERE:      // sphere.zig:55-63
    case TEST_SHAPE:  // shape.zig:429-437
    case CUBE: {      // cube.zig:99-107
      Box b = newBox();
      b.min = point(-1.0, -1.0, -1.0);
      b.max = point(1.0, 1.0, 1.0);
      return b;
    }
    case PLANE: {  // plane.zig:45-53
      Box b = newBox();
      b.min = point(-INF, 0.0, -INF);
      b.max = point(INF, 0.0, INF);
      return b;
    }
    case CYLINDER: {  // cylinder.zig:114-120
      Box b = newBox();
      b.min = point(-1.0, s.ymin, -1.0);
      b.max = point(1.0, s.ymax, 1.0);
      return b;
    }
    case CONE: {  // cone.zig:129-137
      const double limit = std::fmax(std::fabs(s.ymin), std::fabs(s.ymax));
      Box b = newBox();
      b.min = point(-limit, s.ymin, -limit);
      b.max = point(limit, s.ymax, limit);
      return b;
    }
    case TRIANGLE:           // triangle.zig:72-79
    case SMOOTH_TRIANGLE: {  // triangle.zig:267-274
      Box b = newBox();
      boxAdd(b, s.p1);
      boxAdd(b, s.p2);
      boxAdd(b, s.p3);
      return b;
    }
    case GROUP:  // group.zig:81-83
    case CSG:    // csg.zig:108-110
      return groupBox(s);
    default: throw std::runtime_error("oracle scene: bounds of an unsupported shape");
  }
}
inline Box parentSpaceBounds(const Shape& s) { return boxTransform(shapeBounds(s), s.transform); }  // shape.zig:364-370

// ------------------------------------------------------------------ shape.zig constructors that allocate boxes
inline Shape newGroup() {  // shape.zig:235-250: bbox.* = Shape(T).boundingBox() first, then Self.new
  const Box bbox = newBox();
  Shape g = Shape::make(GROUP);
  setGroupBox(g, bbox);
  return g;
}
inline Shape newCsg(Shape left, Shape right, CsgOp op) {  // shape.zig:253-284
  Box bbox = parentSpaceBounds(left);
  boxMerge(bbox, parentSpaceBounds(right));
  Shape c = Shape::make(CSG);
  c.csg_op = op;
  setGroupBox(c, bbox);
  c.children.push_back(std::move(left));
  c.children.push_back(std::move(right));
  return c;
}

// group.zig:75-78
inline void addChild(Shape& group, Shape child) {
  group.children.push_back(std::move(child));
  Box b = groupBox(group);
  boxMerge(b, parentSpaceBounds(group.children.back()));
  setGroupBox(group, b);
}

// shape.zig:286-310
inline void setTransform(Shape& s, const Matrix& m) {
  switch (s.kind) {
    case GROUP: {
      for (Shape& child : s.children) setTransform(child, m.mul(child.transform));
      setGroupBox(s, boxTransform(groupBox(s), m));
      break;
    }
    case CSG:  // (pushes the transform down, never re-boxes)
      setTransform(s.children[0], m.mul(s.children[0].transform));
      setTransform(s.children[1], m.mul(s.children[1].transform));
      break;
    default:
      s.transform = m;
      s.inverse = m.inverse();
      s.inverse_transpose = s.inverse.transpose();
  }
}

// group.zig:85-115
inline void partitionChildren(Shape& g, std::vector<Shape>& left, std::vector<Shape>& right) {
  std::vector<Shape> new_children;
  Box left_box, right_box;
  boxSplit(groupBox(g), left_box, right_box);
  for (Shape& child : g.children) {
    if (boxContainsBox(left_box, parentSpaceBounds(child))) {
      left.push_back(std::move(child));
    } else if (boxContainsBox(right_box, parentSpaceBounds(child))) {
      right.push_back(std::move(child));
    } else {
      new_children.push_back(std::move(child));
    }
  }
  g.children = std::move(new_children);
}
// group.zig:117-135
inline void makeSubgroup(Shape& super, std::vector<Shape>& children) {
  Shape subgroup = newGroup();
  for (Shape& child : children) addChild(subgroup, std::move(child));
  addChild(super, std::move(subgroup));
}
// shape.zig:372-399
inline void divide(Shape& s, size_t threshold) {
  switch (s.kind) {
    case GROUP: {
      if (s.children.size() >= threshold) {
        std::vector<Shape> left, right;
        partitionChildren(s, left, right);
        if (!left.empty()) makeSubgroup(s, left);
        if (!right.empty()) makeSubgroup(s, right);
      }
      for (Shape& child : s.children) divide(child, threshold);
      break;
    }
    case CSG:
      divide(s.children[0], threshold);
      divide(s.children[1], threshold);
      break;
    default: break;
  }
}

// ------------------------------------------------------------------ the built scene
struct Built {
  // patterns, texture maps and images live here: stable addresses (the Pattern tree points into them)
  std::deque<Pattern> patterns;
  std::deque<TextureMap> texmaps;
  std::deque<UvImage> images;
  World world;
  Camera camera;
  size_t first_id = 0;     // the id counter when the parse started
  size_t lines_ignored = 0;  // of the OBJ files (obj.zig:277)
};

// What the caller supplies in place of `load_file_data` (scene.zig:617): the bytes of a named file, and - there is no
// zigimg here - the decoded pixels of a named image (width, height, [h][w][3] colour as zigimg's iterator yields it).
struct Files {
  std::function<std::string(const std::string&)> load;
  std::function<UvImage(const std::string&)> image;
};

// ------------------------------------------------------------------ obj.zig
struct ObjInherited {  // obj.zig:192-195
  const Material* material = nullptr;
  const bool* casts_shadow = nullptr;
};

class ObjParser {  // obj.zig:11-286
 public:
  ObjParser() : default_group_(newGroup()) {}  // obj.zig:32-45

  void loadObj(const std::string& obj, const ObjInherited& state, bool normalize) {  // obj.zig:197-279
    if (normalize) {
      double min_x = INF, min_y = INF, min_z = INF, max_x = -INF, max_y = -INF, max_z = -INF;
      forEachToken(obj, '\n', [&](const std::string& line) {
        std::vector<std::string> tokens = tokenize(line, ' ');
        if (tokens.empty() || tokens[0] != "v") return;
        double x, y, z;
        const bool hx = tokens.size() > 1 && parseFloat(tokens[1], x);
        const bool hy = tokens.size() > 2 && parseFloat(tokens[2], y);
        const bool hz = tokens.size() > 3 && parseFloat(tokens[3], z);
        if (hx) {
          if (x < min_x) min_x = x;
          if (x > max_x) max_x = x;
        }
        if (hy) {
          if (y < min_y) min_y = y;
          if (y > max_y) max_y = y;
        }
        if (hz) {
          if (z < min_z) min_z = z;
          if (z > max_z) max_z = z;
        }
      });
      const double sx = max_x - min_x, sy = max_y - min_y, sz = max_z - min_z;
      const double x_offset = min_x + 0.5 * sx, y_offset = min_y + 0.5 * sy, z_offset = min_z + 0.5 * sz;
      const double scale = 0.5 * std::fmax(sx, std::fmax(sy, sz));
      offset_ = vec3(x_offset, y_offset, z_offset);
      scale_ = scale;
    }
    forEachToken(obj, '\n', [&](const std::string& line) {
      if (!handleLine(line, state)) lines_ignored++;
    });
  }
  Shape toGroup() { return std::move(default_group_); }  // obj.zig:281-283
  size_t lines_ignored = 0;
  // (what the reference's tests look at: parser.vertices / normals / default_group, obj.zig:288-544 - oracle/kat_main.cpp)
  const std::vector<Tuple>& vertices() const { return vertices_; }
  const std::vector<Tuple>& normals() const { return normals_; }
  const Shape& defaultGroup() const { return default_group_; }
  Tuple offset() const { return offset_; }
  double scale() const { return scale_; }

 private:
  Shape default_group_;
  std::vector<size_t> active_path_;  // empty: the default group; {k}: its child k (a named group)
  Tuple offset_ = vec3(0.0, 0.0, 0.0);
  double scale_ = 1.0;
  std::vector<Tuple> vertices_, normals_;

  // std.mem.tokenizeScalar: runs of the delimiter separate tokens, empty tokens do not exist
  template <class F>
  static void forEachToken(const std::string& s, char delim, F&& f) {
    size_t i = 0;
    while (i < s.size()) {
      while (i < s.size() && s[i] == delim) ++i;
      if (i >= s.size()) break;
      size_t j = i;
      while (j < s.size() && s[j] != delim) ++j;
      f(s.substr(i, j - i));
      i = j;
    }
  }
  static std::vector<std::string> tokenize(const std::string& s, char delim) {
    std::vector<std::string> out;
    forEachToken(s, delim, [&](const std::string& t) { out.push_back(t); });
    return out;
  }
  // std.fmt.parseFloat: the whole token must be a number (optional sign, digits, '_' separators are accepted by Zig but
  // do not occur; "inf" / "nan" spellings likewise); strtod with a full-consumption check, hex floats refused
  static bool parseFloat(const std::string& t, double& out) {
    if (t.empty()) return false;
    for (char c : t)
      if (!(std::isdigit(static_cast<unsigned char>(c)) || c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E')) return false;
    char* end = nullptr;
    out = std::strtod(t.c_str(), &end);
    return end == t.c_str() + t.size();
  }
  // std.fmt.parseInt(usize, token, 10): digits only (a leading '+' is accepted by Zig; OBJ files do not have one)
  static bool parseIndex(const std::string& t, size_t& out) {
    if (t.empty()) return false;
    size_t v = 0, i = 0;
    if (t[0] == '+') i = 1;
    if (i >= t.size()) return false;
    for (; i < t.size(); ++i) {
      if (t[i] == '_') continue;
      if (!std::isdigit(static_cast<unsigned char>(t[i]))) return false;
      v = v * 10 + static_cast<size_t>(t[i] - '0');
    }
    out = v;
    return true;
  }
  Shape& active() { return active_path_.empty() ? default_group_ : default_group_.children[active_path_[0]]; }

  bool handleLine(const std::string& line, const ObjInherited& state) {  // obj.zig:172-190; false: the line is ignored
    const std::vector<std::string> tokens = tokenize(line, ' ');
    if (tokens.empty()) return false;  // LineEmpty
    if (tokens[0] == "v") return handleVertex(tokens);
    if (tokens[0] == "vn") return handleVertexNormal(tokens);
    if (tokens[0] == "f") return handleFace(tokens, state);
    if (tokens[0] == "g") return handleNamedGroup(tokens);
    return false;  // UnknownFirstToken
  }
  bool handleVertex(const std::vector<std::string>& tokens) {  // obj.zig:53-67
    double x, y, z;
    if (tokens.size() < 4 || !parseFloat(tokens[1], x) || !parseFloat(tokens[2], y) || !parseFloat(tokens[3], z)) return false;
    vertices_.push_back(div(sub(point(x, y, z), offset_), scale_));
    return true;
  }
  bool handleVertexNormal(const std::vector<std::string>& tokens) {  // obj.zig:69-83
    double x, y, z;
    if (tokens.size() < 4 || !parseFloat(tokens[1], x) || !parseFloat(tokens[2], y) || !parseFloat(tokens[3], z)) return false;
    normals_.push_back(vec3(x, y, z));
    return true;
  }
  struct FaceRef {
    size_t vertex_index = 0;
    bool has_normal = false;
    size_t normal_index = 0;
  };
  static bool faceHelper(const std::string& token, FaceRef& out) {  // obj.zig:85-99 (splitScalar: empty fields exist)
    std::vector<std::string> parts;
    size_t i = 0;
    for (;;) {
      const size_t j = token.find('/', i);
      parts.push_back(token.substr(i, j == std::string::npos ? std::string::npos : j - i));
      if (j == std::string::npos) break;
      i = j + 1;
    }
    if (!parseIndex(parts[0], out.vertex_index)) return false;
    out.has_normal = false;
    if (parts.size() < 3) return true;  // no second '/', or nothing after it
    if (!parseIndex(parts[2], out.normal_index)) return false;  // `try parseInt`: "1/2/" is an error, the line is ignored
    out.has_normal = true;
    return true;
  }
  bool handleFace(const std::vector<std::string>& tokens, const ObjInherited& state) {  // obj.zig:101-150
    if (tokens.size() < 4) return false;  // IncompleteFace (first, last, and at least one more)
    FaceRef first, last;
    if (!faceHelper(tokens[1], first) || !faceHelper(tokens[2], last)) return false;
    for (size_t k = 3; k < tokens.size(); ++k) {
      FaceRef current;
      if (!faceHelper(tokens[k], current)) return false;  // (triangles already added stay, as there)
      // vertices are 1-indexed; an index out of range is a panic in the reference (safety-checked builds)
      if (first.vertex_index - 1 >= vertices_.size() || last.vertex_index - 1 >= vertices_.size() ||
          current.vertex_index - 1 >= vertices_.size())
        throw std::runtime_error("oracle scene: OBJ vertex index out of range");
      const Tuple p1 = vertices_[first.vertex_index - 1];
      const Tuple p2 = vertices_[last.vertex_index - 1];
      const Tuple p3 = vertices_[current.vertex_index - 1];
      auto normal = [&](const FaceRef& f, Tuple& n) {
        if (!f.has_normal) return false;
        if (f.normal_index - 1 >= normals_.size()) throw std::runtime_error("oracle scene: OBJ normal index out of range");
        n = normals_[f.normal_index - 1];
        return true;
      };
      Tuple n1{}, n2{}, n3{};
      const bool h1 = normal(first, n1), h2 = normal(last, n2), h3 = normal(current, n3);
      Shape triangle = (h1 && h2 && h3) ? Shape::smoothTriangle(p1, p2, p3, n1, n2, n3) : Shape::triangle(p1, p2, p3);
      triangle.p2 = p2;
      triangle.p3 = p3;
      triangle.material = state.material ? *state.material : Material{};
      triangle.casts_shadow = state.casts_shadow ? *state.casts_shadow : true;
      addChild(active(), std::move(triangle));
      last = current;
    }
    return true;
  }
  bool handleNamedGroup(const std::vector<std::string>& tokens) {  // obj.zig:152-170
    if (tokens.size() < 2) return false;  // IncompleteNamedGroup
    addChild(default_group_, newGroup());
    active_path_ = {default_group_.children.size() - 1};
    return true;
  }
};

// ------------------------------------------------------------------ scene.zig
struct Inherited {  // scene.zig:432-438 InheritedState
  bool has_material = false;
  Material material;
  Matrix transform = Matrix::identity();
  int casts_shadow = -1;  // ?bool: -1 null
};

class SceneParser {
 public:
  SceneParser(Built& out, const Files& files) : out_(out), files_(files) {}

  void parseScene(const std::string& scene_json, uint32_t width_override, uint32_t height_override) {  // scene.zig:612-661
    const Json root = JsonReader(scene_json).parse();
    out_.first_id = nextId();  // (reads the counter; the parse's first Shape gets first_id + 1)
    if (const Json* defs = root.get("shape-definitions"))
      for (const Json& d : defs->arr) definitions_[d.at("name").str] = &d.at("value");
    const Json& cam = root.at("camera");
    // (F4: the image size is a field of the scene file; the BASELINE configs override it, as the product loader does)
    const size_t width = width_override ? width_override : static_cast<size_t>(cam.at("width").num);
    const size_t height = height_override ? height_override : static_cast<size_t>(cam.at("height").num);
    out_.camera = Camera::make(width, height, cam.at("field-of-view").num);
    const Tuple from = point3(cam.at("from")), to = point3(cam.at("to"));
    const Json& upj = cam.at("up");
    const Tuple up = vec3(upj.arr.at(0).num, upj.arr.at(1).num, upj.arr.at(2).num);
    out_.camera.setTransform(Matrix::viewTransform(from, to, up));
    for (const Json& object : root.at("objects").arr) out_.world.objects.push_back(parseObject(object, Inherited{}));
    for (const Json& light : root.at("lights").arr) {  // scene.zig:593-606
      const Json& l = light.at("point-light");
      const Json& in = l.at("intensity");
      out_.world.lights.push_back({point3(l.at("position")), {in.arr.at(0).num, in.arr.at(1).num, in.arr.at(2).num}});
    }
  }

 private:
  Built& out_;
  const Files& files_;
  std::map<std::string, const Json*> definitions_;

  static Tuple point3(const Json& j) { return point(j.arr.at(0).num, j.arr.at(1).num, j.arr.at(2).num); }
  // a tagged union in std.json's encoding: an object with exactly one member, {"tag": payload}
  static const std::pair<std::string, Json>& tagged(const Json& j) {
    if (j.type != Json::Obj || j.obj.size() != 1) throw std::runtime_error("UnexpectedToken: union");
    return j.obj[0];
  }

  static Matrix parseTransform(const Json& list) {  // scene.zig:214-241
    Matrix matrix = Matrix::identity();
    for (const Json& t : list.arr) {
      const auto& kv = tagged(t);
      const std::string& tag = kv.first;
      const Json& v = kv.second;
      if (tag == "translate") {
        matrix = matrix.translate(v.arr.at(0).num, v.arr.at(1).num, v.arr.at(2).num);
      } else if (tag == "scale") {
        matrix = matrix.scale(v.arr.at(0).num, v.arr.at(1).num, v.arr.at(2).num);
      } else if (tag == "rotate-x") {
        matrix = matrix.rotateX(v.num);
      } else if (tag == "rotate-y") {
        matrix = matrix.rotateY(v.num);
      } else if (tag == "rotate-z") {
        matrix = matrix.rotateZ(v.num);
      } else if (tag == "shear") {  // matrix.zig:299-325: ShearArgs, every field defaults to 0
        auto f = [&](const char* k) { return v.get(k) ? v.get(k)->num : 0.0; };
        matrix = matrix.shear(f("xy"), f("xz"), f("yx"), f("yz"), f("zx"), f("zy"));
      } else {
        throw std::runtime_error("UnknownField: transform " + tag);
      }
    }
    return matrix;
  }

  const Pattern* ownedPattern(const Json& j) {  // `allocator.create(Pattern(T))` + parsePattern
    Pattern p = parsePattern(j);
    out_.patterns.push_back(p);
    return &out_.patterns.back();
  }

  UvPattern parseUvPattern(const Json& j) {  // scene.zig:243-298
    const auto& kv = tagged(j);
    const Json& v = kv.second;
    UvPattern uv;
    if (kv.first == "align-check") {
      uv.kind = UV_ALIGN_CHECK;
      uv.sub[0] = ownedPattern(v.at("central"));
      uv.sub[1] = ownedPattern(v.at("upper-left"));
      uv.sub[2] = ownedPattern(v.at("upper-right"));
      uv.sub[3] = ownedPattern(v.at("bottom-left"));
      uv.sub[4] = ownedPattern(v.at("bottom-right"));
    } else if (kv.first == "checkers") {
      uv.kind = UV_CHECKERS;
      uv.sub[0] = ownedPattern(v.at("patterns").arr.at(0));
      uv.sub[1] = ownedPattern(v.at("patterns").arr.at(1));
      uv.width = v.at("width").num;
      uv.height = v.at("height").num;
    } else if (kv.first == "image") {
      uv.kind = UV_IMAGE;
      out_.images.push_back(files_.image(v.at("file").str));
      uv.image = &out_.images.back();
      const Json* interp = v.get("interpolation");
      uv.bilinear = interp && interp->str == "bilinear";
    } else {
      throw std::runtime_error("UnknownField: uv pattern " + kv.first);
    }
    return uv;
  }

  Pattern parsePattern(const Json& j) {  // scene.zig:300-405
    const auto& kv = tagged(j.at("type"));
    const std::string& tag = kv.first;
    const Json& v = kv.second;
    Pattern pat;
    auto two = [&](PatternKind kind) {
      const Pattern* p1 = ownedPattern(v.arr.at(0));
      const Pattern* p2 = ownedPattern(v.arr.at(1));
      pat.kind = kind;
      pat.a = p1;
      pat.b = p2;
    };
    if (tag == "solid") {
      pat.kind = PAT_SOLID;
      pat.rgb = {v.arr.at(0).num, v.arr.at(1).num, v.arr.at(2).num};
    } else if (tag == "stripes") {
      two(PAT_STRIPES);
    } else if (tag == "rings") {
      two(PAT_RINGS);
    } else if (tag == "gradient") {
      two(PAT_GRADIENT);
    } else if (tag == "radial-gradient") {
      two(PAT_RADIAL_GRADIENT);
    } else if (tag == "checkers") {
      two(PAT_CHECKERS);
    } else if (tag == "blend") {
      two(PAT_BLEND);
    } else if (tag == "perturb") {  // Pattern.perturb(p1, .{}): PerturbInfo's defaults, perturb.zig:21-25
      pat.kind = PAT_PERTURB;
      pat.a = ownedPattern(v);
      pat.rgb = {0.3, 3.0, 0.8};
    } else if (tag == "texture-map") {
      const auto& mk = tagged(v);
      TextureMap tm;
      if (mk.first == "spherical") {
        tm.mapping = TEX_SPHERICAL;
        tm.faces[0] = parseUvPattern(mk.second.at("uv-pattern"));
      } else if (mk.first == "planar") {
        tm.mapping = TEX_PLANAR;
        tm.faces[0] = parseUvPattern(mk.second.at("uv-pattern"));
      } else if (mk.first == "cylindrical") {
        tm.mapping = TEX_CYLINDRICAL;
        tm.faces[0] = parseUvPattern(mk.second.at("uv-pattern"));
      } else if (mk.first == "cubic") {
        tm.mapping = TEX_CUBIC;
        tm.faces[0] = parseUvPattern(mk.second.at("front"));
        tm.faces[1] = parseUvPattern(mk.second.at("back"));
        tm.faces[2] = parseUvPattern(mk.second.at("left"));
        tm.faces[3] = parseUvPattern(mk.second.at("right"));
        tm.faces[4] = parseUvPattern(mk.second.at("up"));
        tm.faces[5] = parseUvPattern(mk.second.at("down"));
      } else {
        throw std::runtime_error("UnknownMapping");
      }
      out_.texmaps.push_back(tm);
      pat.kind = PAT_TEXTURE_MAP;
      pat.texture_map = &out_.texmaps.back();
    } else {
      throw std::runtime_error("UnknownField: pattern " + tag);
    }
    if (const Json* t = j.get("transform"))
      if (t->type != Json::Null) pat.setTransform(parseTransform(*t));
    return pat;
  }

  Material parseMaterial(const Json& j, const Inherited& inherited) {  // scene.zig:407-430
    Material mat = inherited.has_material ? inherited.material : Material{};
    if (const Json* p = j.get("pattern"))
      if (p->type != Json::Null) mat.pattern = parsePattern(*p);
    auto opt = [&](const char* k, double& field) {
      const Json* v = j.get(k);
      if (v && v->type != Json::Null) field = v->num;
    };
    opt("ambient", mat.ambient);
    opt("diffuse", mat.diffuse);
    opt("specular", mat.specular);
    opt("shininess", mat.shininess);
    opt("reflective", mat.reflective);
    opt("transparency", mat.transparency);
    opt("refractive-index", mat.refractive_index);
    return mat;
  }

  Inherited inherit(const Json& object, const Inherited& inherited) {  // scene.zig:164-190
    Inherited info = inherited;
    const Json* m = object.get("material");
    if (m && m->type != Json::Null) {
      info.material = parseMaterial(*m, inherited);
      info.has_material = true;
    }
    const Json* t = object.get("transform");
    if (t && t->type != Json::Null) info.transform = parseTransform(*t).mul(inherited.transform);
    const Json* s = object.get("casts-shadow");
    if (s && s->type != Json::Null) info.casts_shadow = s->b ? 1 : 0;
    return info;
  }

  Shape parseObject(const Json& object, const Inherited& inherited) {  // scene.zig:440-591
    const Inherited info = inherit(object, inherited);
    bool has_material = info.has_material;
    Material material = info.material;
    Matrix transform = info.transform;
    int casts_shadow = info.casts_shadow;

    const auto& kv = tagged(object.at("type"));
    const std::string& tag = kv.first;
    const Json& v = kv.second;
    Shape shape;
    auto childState = [&]() {  // .{ .material = material, .casts_shadow = casts_shadow }: the transform is NOT passed down
      Inherited st;
      st.has_material = has_material;
      st.material = material;
      st.casts_shadow = casts_shadow;
      return st;
    };
    if (tag == "from-definition") {
      const auto def = definitions_.find(v.str);
      if (def == definitions_.end()) throw std::runtime_error("UnknownDefinition");
      Inherited st = childState();
      st.transform = inherited.transform;
      Shape parent = parseObject(*def->second, st);
      Inherited parent_state;
      parent_state.has_material = true;  // parent.material is a Material, not an optional
      parent_state.material = parent.material;
      parent_state.transform = parent.transform;
      parent_state.casts_shadow = parent.casts_shadow ? 1 : 0;
      const Inherited again = inherit(object, parent_state);
      has_material = again.has_material;
      material = again.material;
      transform = again.transform;
      casts_shadow = again.casts_shadow;
      shape = std::move(parent);
    } else if (tag == "from-obj") {
      const std::string obj = files_.load(v.at("file").str);
      ObjParser parser;
      ObjInherited st;
      if (has_material) st.material = &material;
      const bool shadow = casts_shadow == 1;
      if (casts_shadow >= 0) st.casts_shadow = &shadow;
      const Json* norm = v.get("normalize");
      parser.loadObj(obj, st, norm ? norm->b : true);
      out_.lines_ignored += parser.lines_ignored;
      shape = parser.toGroup();
    } else if (tag == "sphere") {
      shape = Shape::make(SPHERE);
    } else if (tag == "plane") {
      shape = Shape::make(PLANE);
    } else if (tag == "cube") {
      shape = Shape::make(CUBE);
    } else if (tag == "cylinder" || tag == "cone") {
      shape = Shape::make(tag == "cone" ? CONE : CYLINDER);
      if (const Json* f = v.get("min")) shape.ymin = f->num;
      if (const Json* f = v.get("max")) shape.ymax = f->num;
      if (const Json* f = v.get("closed")) shape.closed = f->b;
    } else if (tag == "triangle") {
      const Tuple p1 = point3(v.at("p1")), p2 = point3(v.at("p2")), p3 = point3(v.at("p3"));
      shape = Shape::triangle(p1, p2, p3);
      shape.p2 = p2;
      shape.p3 = p3;
    } else if (tag == "group") {
      Shape g = newGroup();
      for (const Json& child : v.arr) addChild(g, parseObject(child, childState()));
      shape = std::move(g);
    } else if (tag == "csg") {
      Shape left = parseObject(v.at("left"), childState());
      Shape right = parseObject(v.at("right"), childState());
      const std::string& opname = v.at("operation").str;
      const CsgOp op = opname == "union" ? CSG_UNION : (opname == "intersection" ? CSG_INTERSECTION : CSG_DIFFERENCE);
      if (opname != "union" && opname != "intersection" && opname != "difference") throw std::runtime_error("InvalidEnumTag");
      shape = newCsg(std::move(left), std::move(right), op);
    } else {
      throw std::runtime_error("UnknownField: shape " + tag);
    }

    setTransform(shape, transform);
    if (has_material) shape.material = material;
    if (casts_shadow >= 0) shape.casts_shadow = casts_shadow == 1;
    divide(shape, 8);
    return shape;
  }
};

inline std::unique_ptr<Built> buildScene(const std::string& scene_json, const Files& files, uint32_t width = 0, uint32_t height = 0) {
  auto out = std::make_unique<Built>();
  SceneParser(*out, files).parseScene(scene_json, width, height);
  return out;
}

}  // namespace scene
}  // namespace orc
