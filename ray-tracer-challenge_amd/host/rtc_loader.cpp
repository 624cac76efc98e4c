// rtc_loader.cpp — scene JSON + OBJ loading.
//
// Restated from the reference (quirks listed in SURVEY §3.3 are kept on purpose:
// they fix leaf matrices, leaf order and group boxes, i.e. the kernel's input):
//   scene.zig:164-190   ObjectConfig.inherit (own transform applied AFTER the inherited one)
//   scene.zig:214-241   parseTransform (list order, each op left-multiplies)
//   scene.zig:300-405   parsePattern
//   scene.zig:407-430   parseMaterial (override-merge on the inherited material)
//   scene.zig:440-591   parseObject (from-definition double inherit, group children
//                       parsed with identity transform, setTransform, divide(8))
//   scene.zig:593-661   parseLight, parseScene
//   obj.zig:53-283      ObjParser
#include "rtc_loader.hpp"

#include <sched.h>
#include <zlib.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <exception>
#include <system_error>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "rtc_json.hpp"

namespace rtc {

using json::Value;

FileLoader directoryLoader(const std::string& dir) {
  return [dir](const std::string& name) {
    const std::string path = dir.empty() ? name : dir + "/" + name;
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("FileNotFound", path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
  };
}

namespace {

// ---- typed access with std.json-style errors -------------------------------------------
const Value& requireObject(const Value& v, const char* what) {
  if (v.type != Value::Object) throw Error("UnexpectedToken", std::string("expected object for ") + what);
  return v;
}
const Value& requireArray(const Value& v, const char* what) {
  if (v.type != Value::Array) throw Error("UnexpectedToken", std::string("expected array for ") + what);
  return v;
}
double asFloat(const Value& v, const char* what) {
  if (v.type != Value::Number) throw Error("UnexpectedToken", std::string("expected number for ") + what);
  return v.num;
}
size_t asUsize(const Value& v, const char* what) {
  if (v.type != Value::Number) throw Error("UnexpectedToken", std::string("expected integer for ") + what);
  if (v.num < 0) throw Error("Overflow", what);
  const double r = std::floor(v.num);
  if (r != v.num) throw Error("InvalidNumber", what);
  return static_cast<size_t>(r);
}
bool asBool(const Value& v, const char* what) {
  if (v.type != Value::Bool) throw Error("UnexpectedToken", std::string("expected bool for ") + what);
  return v.b;
}
const std::string& asString(const Value& v, const char* what) {
  if (v.type != Value::String) throw Error("UnexpectedToken", std::string("expected string for ") + what);
  return v.str;
}
void asVec3(const Value& v, const char* what, double out[3]) {
  requireArray(v, what);
  if (v.arr.size() != 3) throw Error("LengthMismatch", what);
  for (int i = 0; i < 3; ++i) out[i] = asFloat(v.arr[i], what);
}
// Checks an object against its allowed field list: duplicates and unknown names are
// errors (std.json defaults: duplicate_field_behavior=.error, ignore_unknown_fields=false).
void checkFields(const Value& obj, std::initializer_list<const char*> allowed, const char* what) {
  for (size_t i = 0; i < obj.obj.size(); ++i) {
    const std::string& k = obj.obj[i].first;
    bool ok = false;
    for (const char* a : allowed) ok = ok || k == a;
    if (!ok) throw Error("UnknownField", std::string(what) + "." + k);
    for (size_t j = 0; j < i; ++j)
      if (obj.obj[j].first == k) throw Error("DuplicateField", std::string(what) + "." + k);
  }
}
const Value& requireField(const Value& obj, const char* key, const char* what) {
  const Value* v = obj.find(key);
  if (!v) throw Error("MissingField", std::string(what) + "." + key);
  return *v;
}
// A Zig union(enum) is encoded as an object with exactly one member.
const std::pair<std::string, Value>& unionMember(const Value& v, const char* what) {
  requireObject(v, what);
  if (v.obj.size() != 1) throw Error("UnexpectedToken", std::string("union ") + what + " needs exactly one member");
  return v.obj[0];
}
void requireVoid(const Value& v, const char* what) {  // `sphere: void` is written {}
  if (v.type != Value::Object || !v.obj.empty()) throw Error("UnexpectedToken", std::string("expected {} for ") + what);
}

// ---- scene.zig:214-241 ------------------------------------------------------------------
Matrix4 parseTransform(const Value& list) {
  requireArray(list, "transform");
  Matrix4 m = Matrix4::identity();
  for (const Value& item : list.arr) {
    const auto& kv = unionMember(item, "transform entry");
    const std::string& op = kv.first;
    if (op == "translate" || op == "scale") {
      double b[3];
      asVec3(kv.second, op.c_str(), b);
      m = (op == "translate") ? m.translate(b[0], b[1], b[2]) : m.scale(b[0], b[1], b[2]);
    } else if (op == "rotate-x") {
      m = m.rotateX(asFloat(kv.second, "rotate-x"));
    } else if (op == "rotate-y") {
      m = m.rotateY(asFloat(kv.second, "rotate-y"));
    } else if (op == "rotate-z") {
      m = m.rotateZ(asFloat(kv.second, "rotate-z"));
    } else if (op == "shear") {
      const Value& o = requireObject(kv.second, "shear");
      checkFields(o, {"xy", "xz", "yx", "yz", "zx", "zy"}, "shear");
      ShearArgs a;
      if (auto* v = o.find("xy")) a.xy = asFloat(*v, "xy");
      if (auto* v = o.find("xz")) a.xz = asFloat(*v, "xz");
      if (auto* v = o.find("yx")) a.yx = asFloat(*v, "yx");
      if (auto* v = o.find("yz")) a.yz = asFloat(*v, "yz");
      if (auto* v = o.find("zx")) a.zx = asFloat(*v, "zx");
      if (auto* v = o.find("zy")) a.zy = asFloat(*v, "zy");
      m = m.shear(a);
    } else {
      throw Error("UnknownField", "transform." + op);
    }
  }
  return m;
}

// ---- scene.zig:300-405 ------------------------------------------------------------------
// A PNG as zigimg hands it to Canvas.fromImage (scene.zig:282-285, canvas.zig:34-46): every pixel as an f32
// colour, channel / max (zigimg color.zig toF32Color; the pinned zigimg is not in the tree).  Supports what
// the reference's data files are and a little more: non-interlaced, 8 or 16 bits, grey / RGB / palette /
// with alpha (dropped: canvas.zig:41 keeps r, g, b).
UvImageData decodePng(const std::string& bytes) {
  auto be32 = [&](size_t off) {
    return (static_cast<uint32_t>(static_cast<unsigned char>(bytes[off])) << 24) |
           (static_cast<uint32_t>(static_cast<unsigned char>(bytes[off + 1])) << 16) |
           (static_cast<uint32_t>(static_cast<unsigned char>(bytes[off + 2])) << 8) |
           static_cast<uint32_t>(static_cast<unsigned char>(bytes[off + 3]));
  };
  static const unsigned char kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (bytes.size() < 8 + 25 || std::memcmp(bytes.data(), kSig, 8) != 0)
    throw Error("Unsupported", "image: not a PNG (the reference decodes other formats through zigimg)");
  uint32_t width = 0, height = 0;
  unsigned depth = 0, color = 0, interlace = 0;
  std::string idat;
  std::vector<unsigned char> palette;
  size_t off = 8;
  bool seen_ihdr = false, seen_iend = false;
  while (off + 12 <= bytes.size() && !seen_iend) {
    const uint32_t len = be32(off);
    const std::string type = bytes.substr(off + 4, 4);
    if (off + 12 + static_cast<size_t>(len) > bytes.size()) throw Error("EndOfStream", "png chunk " + type);
    const size_t data = off + 8;
    if (type == "IHDR") {
      if (len != 13) throw Error("InvalidData", "png IHDR");
      width = be32(data);
      height = be32(data + 4);
      depth = static_cast<unsigned char>(bytes[data + 8]);
      color = static_cast<unsigned char>(bytes[data + 9]);
      interlace = static_cast<unsigned char>(bytes[data + 12]);
      seen_ihdr = true;
    } else if (type == "PLTE") {
      palette.assign(bytes.begin() + data, bytes.begin() + data + len);
    } else if (type == "IDAT") {
      idat.append(bytes, data, len);
    } else if (type == "IEND") {
      seen_iend = true;
    }
    off += 12 + static_cast<size_t>(len);
  }
  if (!seen_ihdr || width == 0 || height == 0) throw Error("InvalidData", "png without IHDR");
  if (interlace != 0) throw Error("Unsupported", "interlaced png");
  if (depth != 8 && depth != 16) throw Error("Unsupported", "png bit depth " + std::to_string(depth));
  unsigned channels = 0;
  switch (color) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw Error("InvalidData", "png colour type");
  }
  if (color == 3 && depth != 8) throw Error("Unsupported", "png palette depth");
  const size_t bpp = channels * (depth / 8), stride = static_cast<size_t>(width) * bpp;
  std::vector<unsigned char> raw((stride + 1) * height);
  uLongf out_len = static_cast<uLongf>(raw.size());
  if (uncompress(raw.data(), &out_len, reinterpret_cast<const Bytef*>(idat.data()), static_cast<uLong>(idat.size())) != Z_OK ||
      out_len != raw.size())
    throw Error("InvalidData", "png IDAT does not inflate to the image size");
  // undo the scanline filters (PNG spec 9.2)
  std::vector<unsigned char> img(stride * height);
  for (uint32_t y = 0; y < height; ++y) {
    const unsigned char filter = raw[(stride + 1) * y];
    const unsigned char* in = &raw[(stride + 1) * y + 1];
    unsigned char* cur = &img[stride * y];
    const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int pred = 0;
      switch (filter) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) / 2; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: throw Error("InvalidData", "png filter type");
      }
      cur[i] = static_cast<unsigned char>(in[i] + pred);
    }
  }
  UvImageData out;
  out.width = width;
  out.height = height;
  out.rgb.resize(static_cast<size_t>(width) * height * 3);
  auto sample = [&](const unsigned char* p) {  // one channel as zigimg's toF32Color
    return depth == 8 ? static_cast<float>(p[0]) / 255.0f : static_cast<float>((p[0] << 8) | p[1]) / 65535.0f;
  };
  for (size_t i = 0; i < static_cast<size_t>(width) * height; ++i) {
    const unsigned char* px = &img[i * bpp];
    float r, g, b;
    if (color == 3) {
      const size_t k = px[0];
      if (3 * k + 2 >= palette.size()) throw Error("InvalidData", "png palette index");
      r = static_cast<float>(palette[3 * k]) / 255.0f;
      g = static_cast<float>(palette[3 * k + 1]) / 255.0f;
      b = static_cast<float>(palette[3 * k + 2]) / 255.0f;
    } else if (channels <= 2) {
      r = g = b = sample(px);
    } else {
      r = sample(px);
      g = sample(px + depth / 8);
      b = sample(px + 2 * (depth / 8));
    }
    out.rgb[3 * i] = r;
    out.rgb[3 * i + 1] = g;
    out.rgb[3 * i + 2] = b;
  }
  return out;
}

Pattern parsePattern(const Value& cfg, const FileLoader& load_file_data);

// scene.zig:243-298
UvPattern parseUvPattern(const Value& cfg, const FileLoader& load_file_data) {
  const auto& kv = unionMember(cfg, "uv-pattern");
  const std::string& t = kv.first;
  UvPattern uv;
  auto sub = [&](const Value& v) { return std::make_shared<const Pattern>(parsePattern(v, load_file_data)); };
  if (t == "align-check") {
    requireObject(kv.second, "align-check");
    checkFields(kv.second, {"central", "upper-left", "upper-right", "bottom-left", "bottom-right"}, "align-check");
    uv.kind = UvKind::AlignCheck;
    for (const char* f : {"central", "upper-left", "upper-right", "bottom-left", "bottom-right"})
      uv.sub.push_back(sub(requireField(kv.second, f, "align-check")));
  } else if (t == "checkers") {
    requireObject(kv.second, "uv checkers");
    checkFields(kv.second, {"width", "height", "patterns"}, "uv checkers");
    uv.kind = UvKind::Checkers;
    uv.width = asFloat(requireField(kv.second, "width", "uv checkers"), "width");
    uv.height = asFloat(requireField(kv.second, "height", "uv checkers"), "height");
    const Value& ps = requireField(kv.second, "patterns", "uv checkers");
    requireArray(ps, "patterns");
    if (ps.arr.size() != 2) throw Error("LengthMismatch", "uv checkers patterns");
    uv.sub.push_back(sub(ps.arr[0]));
    uv.sub.push_back(sub(ps.arr[1]));
  } else if (t == "image") {
    requireObject(kv.second, "image");
    checkFields(kv.second, {"file", "interpolation"}, "image");
    uv.kind = UvKind::Image;
    const std::string& file = asString(requireField(kv.second, "file", "image"), "file");
    if (const Value* ip = kv.second.find("interpolation")) {
      const std::string& mode = asString(*ip, "interpolation");
      if (mode == "bilinear") {
        uv.bilinear = true;
      } else if (mode != "none") {
        throw Error("InvalidEnumTag", "image.interpolation " + mode);
      }
    }
    uv.image = std::make_shared<const UvImageData>(decodePng(load_file_data(file)));
  } else {
    throw Error("UnknownField", "uv-pattern." + t);
  }
  return uv;
}

Pattern parsePattern(const Value& cfg, const FileLoader& load_file_data) {
  requireObject(cfg, "pattern");
  checkFields(cfg, {"type", "transform"}, "pattern");
  const auto& kv = unionMember(requireField(cfg, "type", "pattern"), "pattern.type");
  const std::string& t = kv.first;
  Pattern pat;
  auto two = [&](PatternKind k) {
    requireArray(kv.second, t.c_str());
    if (kv.second.arr.size() != 2) throw Error("LengthMismatch", t);
    return Pattern::binary(k, parsePattern(kv.second.arr[0], load_file_data), parsePattern(kv.second.arr[1], load_file_data));
  };
  if (t == "solid") {
    double c[3];
    asVec3(kv.second, "solid", c);
    pat = Pattern::solid({c[0], c[1], c[2]});
  } else if (t == "stripes") {
    pat = two(PatternKind::Stripes);
  } else if (t == "rings") {
    pat = two(PatternKind::Rings);
  } else if (t == "gradient") {
    pat = two(PatternKind::Gradient);
  } else if (t == "radial-gradient") {
    pat = two(PatternKind::RadialGradient);
  } else if (t == "checkers") {
    pat = two(PatternKind::Checkers);
  } else if (t == "blend") {
    pat = two(PatternKind::Blend);
  } else if (t == "perturb") {
    pat.kind = PatternKind::Perturb;
    pat.a = std::make_shared<const Pattern>(parsePattern(kv.second, load_file_data));
  } else if (t == "texture-map") {  // scene.zig:366-396
    const auto& mk = unionMember(kv.second, "texture-map");
    TextureMap tm;
    auto single = [&](TexMapping m) {
      requireObject(mk.second, mk.first.c_str());
      checkFields(mk.second, {"uv-pattern"}, mk.first.c_str());
      tm.mapping = m;
      tm.faces.push_back(parseUvPattern(requireField(mk.second, "uv-pattern", mk.first.c_str()), load_file_data));
    };
    if (mk.first == "spherical") {
      single(TexMapping::Spherical);
    } else if (mk.first == "planar") {
      single(TexMapping::Planar);
    } else if (mk.first == "cylindrical") {
      single(TexMapping::Cylindrical);
    } else if (mk.first == "cubic") {
      requireObject(mk.second, "cubic");
      checkFields(mk.second, {"front", "back", "left", "right", "up", "down"}, "cubic");
      tm.mapping = TexMapping::Cubic;
      for (const char* f : {"front", "back", "left", "right", "up", "down"})  // scene.zig:387-393, Cubic.Face order
        tm.faces.push_back(parseUvPattern(requireField(mk.second, f, "cubic"), load_file_data));
    } else {
      throw Error("UnknownField", "texture-map." + mk.first);
    }
    pat.kind = PatternKind::TextureMap;
    pat.texture_map = std::make_shared<const TextureMap>(std::move(tm));
  } else {
    throw Error("UnknownField", "pattern.type." + t);
  }
  if (const Value* tr = cfg.find("transform")) {
    if (tr->type != Value::Null) pat.setTransform(parseTransform(*tr));
  }
  return pat;
}

// ---- scene.zig:407-430 ------------------------------------------------------------------
Material parseMaterial(const Value& cfg, const std::optional<Material>& inherited, const FileLoader& load_file_data) {
  requireObject(cfg, "material");
  checkFields(cfg, {"pattern", "ambient", "diffuse", "specular", "shininess", "reflective", "transparency",
                    "refractive-index"}, "material");
  Material mat = inherited ? *inherited : Material{};
  auto present = [&](const char* k) -> const Value* {
    const Value* v = cfg.find(k);
    return (v && v->type != Value::Null) ? v : nullptr;
  };
  if (auto* v = present("pattern")) mat.pattern = parsePattern(*v, load_file_data);
  if (auto* v = present("ambient")) mat.ambient = asFloat(*v, "ambient");
  if (auto* v = present("diffuse")) mat.diffuse = asFloat(*v, "diffuse");
  if (auto* v = present("specular")) mat.specular = asFloat(*v, "specular");
  if (auto* v = present("shininess")) mat.shininess = asFloat(*v, "shininess");
  if (auto* v = present("reflective")) mat.reflective = asFloat(*v, "reflective");
  if (auto* v = present("transparency")) mat.transparency = asFloat(*v, "transparency");
  if (auto* v = present("refractive-index")) mat.refractive_index = asFloat(*v, "refractive-index");
  return mat;
}

struct InheritedState {  // scene.zig:432-438
  std::optional<Material> material;
  Matrix4 transform = Matrix4::identity();
  std::optional<bool> casts_shadow;
};

struct Info {
  std::optional<Material> material;
  Matrix4 transform;
  std::optional<bool> casts_shadow;
};

const Value* presentField(const Value& obj, const char* k) {
  const Value* v = obj.find(k);
  return (v && v->type != Value::Null) ? v : nullptr;
}

// scene.zig:164-190
Info inherit(const Value& object, const InheritedState& inherited, const FileLoader& load_file_data) {
  Info info;
  if (const Value* m = presentField(object, "material")) {
    info.material = parseMaterial(*m, inherited.material, load_file_data);
  } else {
    info.material = inherited.material;
  }
  if (const Value* t = presentField(object, "transform")) {
    info.transform = parseTransform(*t).mul(inherited.transform);
  } else {
    info.transform = inherited.transform;
  }
  if (const Value* s = presentField(object, "casts-shadow")) {
    info.casts_shadow = asBool(*s, "casts-shadow");
  } else {
    info.casts_shadow = inherited.casts_shadow;
  }
  return info;
}

using Definitions = std::map<std::string, const Value*>;

void parseMinMaxClosed(const Value& cfg, const char* what, Shape& s) {  // scene.zig:134-143
  requireObject(cfg, what);
  checkFields(cfg, {"min", "max", "closed"}, what);
  if (auto* v = cfg.find("min")) s.ymin = asFloat(*v, "min");
  if (auto* v = cfg.find("max")) s.ymax = asFloat(*v, "max");
  if (auto* v = cfg.find("closed")) s.closed = asBool(*v, "closed");
}

// scene.zig:440-591
Shape parseObject(const Value& object, const InheritedState& inherited, const Definitions& definitions,
                  const FileLoader& load_file_data, int depth = 0) {
  if (depth > 64) throw Error("StackOverflow", "shape definitions nest too deeply (cycle?)");
  requireObject(object, "object");
  checkFields(object, {"type", "transform", "material", "casts-shadow"}, "object");

  const Info info = inherit(object, inherited, load_file_data);
  std::optional<Material> material = info.material;
  Matrix4 transform = info.transform;
  std::optional<bool> casts_shadow = info.casts_shadow;

  const auto& kv = unionMember(requireField(object, "type", "object"), "object.type");
  const std::string& t = kv.first;
  const Value& payload = kv.second;

  Shape shape;
  if (t == "from-definition") {
    const std::string& name = asString(payload, "from-definition");
    auto it = definitions.find(name);
    if (it == definitions.end()) throw Error("UnknownDefinition", name);
    // scene.zig:455-492: the definition is parsed with THIS object's merged material and
    // shadow flag but only the inherited transform; then this object's own fields are
    // inherited again on top of what the parsed definition ended up with.
    InheritedState for_def;
    for_def.material = material;
    for_def.transform = inherited.transform;
    for_def.casts_shadow = casts_shadow;
    Shape parent = parseObject(*it->second, for_def, definitions, load_file_data, depth + 1);
    InheritedState parent_state;
    parent_state.material = parent.material;
    parent_state.transform = parent.transform;
    parent_state.casts_shadow = parent.casts_shadow;
    const Info again = inherit(object, parent_state, load_file_data);
    material = again.material;
    transform = again.transform;
    casts_shadow = again.casts_shadow;
    shape = std::move(parent);
  } else if (t == "from-obj") {
    const Value& cfg = requireObject(payload, "from-obj");
    checkFields(cfg, {"file", "normalize"}, "from-obj");
    const std::string& file = asString(requireField(cfg, "file", "from-obj"), "file");
    bool normalize = true;  // scene.zig:124-127
    if (auto* v = cfg.find("normalize")) normalize = asBool(*v, "normalize");
    const std::string obj = load_file_data(file);
    ObjParser parser;
    ObjParser::InheritedState st;
    st.material = material;
    st.casts_shadow = casts_shadow;
    parser.loadObj(obj, st, normalize);
    shape = parser.takeGroup();
  } else if (t == "sphere") {
    requireVoid(payload, "sphere");
    shape = Shape::sphere();
  } else if (t == "plane") {
    requireVoid(payload, "plane");
    shape = Shape::plane();
  } else if (t == "cube") {
    requireVoid(payload, "cube");
    shape = Shape::cube();
  } else if (t == "cylinder") {
    shape = Shape::cylinder();
    parseMinMaxClosed(payload, "cylinder", shape);
  } else if (t == "cone") {
    shape = Shape::cone();
    parseMinMaxClosed(payload, "cone", shape);
  } else if (t == "triangle") {
    const Value& cfg = requireObject(payload, "triangle");
    checkFields(cfg, {"p1", "p2", "p3"}, "triangle");
    double a[3], b[3], c[3];
    asVec3(requireField(cfg, "p1", "triangle"), "p1", a);
    asVec3(requireField(cfg, "p2", "triangle"), "p2", b);
    asVec3(requireField(cfg, "p3", "triangle"), "p3", c);
    shape = Shape::triangle(Tuple::point(a[0], a[1], a[2]), Tuple::point(b[0], b[1], b[2]),
                            Tuple::point(c[0], c[1], c[2]));
  } else if (t == "group") {
    requireArray(payload, "group");
    shape = Shape::group();
    for (const Value& child : payload.arr) {
      // Groups push their own transform down in setTransform; children are parsed with
      // the inherited material / shadow flag but an identity transform (scene.zig:527-546).
      InheritedState st;
      st.material = material;
      st.casts_shadow = casts_shadow;
      shape.addChild(parseObject(child, st, definitions, load_file_data, depth + 1));
    }
  } else if (t == "csg") {  // scene.zig:547-575
    requireObject(payload, "csg");
    checkFields(payload, {"left", "right", "operation"}, "csg");
    InheritedState st;
    st.material = material;
    st.casts_shadow = casts_shadow;
    Shape left = parseObject(requireField(payload, "left", "csg"), st, definitions, load_file_data, depth + 1);
    Shape right = parseObject(requireField(payload, "right", "csg"), st, definitions, load_file_data, depth + 1);
    const std::string& op = asString(requireField(payload, "operation", "csg"), "operation");
    CsgOp o;
    if (op == "union") {
      o = CsgOp::Union;
    } else if (op == "intersection") {
      o = CsgOp::Intersection;
    } else if (op == "difference") {
      o = CsgOp::Difference;
    } else {
      throw Error("InvalidEnumTag", "csg.operation " + op);
    }
    shape = Shape::csg(std::move(left), std::move(right), o);
  } else {
    throw Error("UnknownField", "object.type." + t);
  }

  shape.setTransform(transform);           // scene.zig:578
  if (material) shape.material = *material;  // scene.zig:580-582
  if (casts_shadow) shape.casts_shadow = *casts_shadow;
  shape.divide(8);                         // scene.zig:588
  return shape;
}

}  // namespace

namespace {
std::atomic<unsigned> g_loader_threads{0};
}
void setLoaderThreads(unsigned threads) { g_loader_threads.store(threads); }
unsigned loaderThreads() {
  const unsigned asked = g_loader_threads.load();
  if (asked) return asked;
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, std::max(1, CPU_COUNT(&set)));
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
    long long quota = 0, period = 0;
    if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
      n = std::min<unsigned>(n, static_cast<unsigned>(std::max<long long>(1, quota / period)));
    std::fclose(f);
  }
  return std::min(16u, n);
}

// ---- scene.zig:612-661 ------------------------------------------------------------------
SceneInfo parseScene(const std::string& scene_json, const FileLoader& load_file_data) {
  const Value root = json::parse(scene_json);
  requireObject(root, "scene");
  checkFields(root, {"shape-definitions", "camera", "lights", "objects"}, "scene");

  Definitions definitions;
  if (const Value* defs = root.find("shape-definitions")) {
    requireArray(*defs, "shape-definitions");
    for (const Value& d : defs->arr) {
      requireObject(d, "shape-definition");
      checkFields(d, {"name", "value"}, "shape-definition");
      const std::string& name = asString(requireField(d, "name", "shape-definition"), "name");
      definitions[name] = &requireField(d, "value", "shape-definition");  // put(): last one wins
    }
  }

  const Value& cam = requireObject(requireField(root, "camera", "scene"), "camera");
  checkFields(cam, {"width", "height", "field-of-view", "from", "to", "up"}, "camera");
  SceneInfo info;
  info.camera = Camera::create(asUsize(requireField(cam, "width", "camera"), "width"),
                               asUsize(requireField(cam, "height", "camera"), "height"),
                               asFloat(requireField(cam, "field-of-view", "camera"), "field-of-view"));
  double f[3], t[3], u[3];
  asVec3(requireField(cam, "from", "camera"), "from", f);
  asVec3(requireField(cam, "to", "camera"), "to", t);
  asVec3(requireField(cam, "up", "camera"), "up", u);
  const Tuple from = Tuple::point(f[0], f[1], f[2]);
  const Tuple to = Tuple::point(t[0], t[1], t[2]);
  const Tuple up = Tuple::vec3(u[0], u[1], u[2]);
  info.camera.saved_from = from;
  info.camera.saved_to = to;
  info.camera.saved_up = up;
  info.camera.setTransform(Matrix4::viewTransform(from, to, up));

  const Value& objects = requireArray(requireField(root, "objects", "scene"), "objects");
  // `lights` has no default in SceneConfig (scene.zig:206): it is required.
  const Value& lights = requireArray(requireField(root, "lights", "scene"), "lights");

  // scene.zig:650-655 is one loop; here the objects are built side by side and numbered afterwards, in their order, with
  // the ids that loop would have drawn (dragons.json: six dragons of 23 490 triangles each, 0.5 s of parsing and dividing)
  const size_t n_objects = objects.arr.size();
  const size_t n_threads = std::min<size_t>(n_objects, loaderThreads());
  if (n_threads <= 1) {
    for (const Value& o : objects.arr)
      info.world.objects.push_back(parseObject(o, InheritedState{}, definitions, load_file_data));
  } else {
    std::vector<Shape> built(n_objects);
    std::vector<size_t> ids_drawn(n_objects, 0);
    std::vector<std::exception_ptr> failed(n_objects);
    std::atomic<size_t> next{0};
    auto work = [&] {
      for (size_t i = next.fetch_add(1); i < n_objects; i = next.fetch_add(1)) {
        ShapeIdScope ids;
        try {
          built[i] = parseObject(objects.arr[i], InheritedState{}, definitions, load_file_data);
        } catch (...) {
          failed[i] = std::current_exception();
        }
        ids_drawn[i] = ids.drawn;
      }
    };
    std::vector<std::thread> pool;
    pool.reserve(n_threads);
    try {
      for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
    } catch (const std::system_error&) {
      // (no more threads to be had: the ones that started and this one share the work)
    }
    work();
    for (std::thread& t : pool) t.join();
    for (size_t i = 0; i < n_objects; ++i) {
      if (failed[i]) std::rethrow_exception(failed[i]);  // (the first object in the file's order that fails, as in the one loop)
      offsetShapeIds(built[i], reserveShapeIds(ids_drawn[i]));
      info.world.objects.push_back(std::move(built[i]));
    }
  }

  for (const Value& l : lights.arr) {  // scene.zig:593-606
    const auto& kv = unionMember(l, "light");
    if (kv.first != "point-light") throw Error("UnknownField", "light." + kv.first);
    const Value& cfg = requireObject(kv.second, "point-light");
    checkFields(cfg, {"position", "intensity"}, "point-light");
    double p[3], c[3];
    asVec3(requireField(cfg, "position", "point-light"), "position", p);
    asVec3(requireField(cfg, "intensity", "point-light"), "intensity", c);
    info.world.lights.push_back({Tuple::point(p[0], p[1], p[2]), {c[0], c[1], c[2]}});
  }
  return info;
}

// =========================================================================================
// OBJ parser, obj.zig
// =========================================================================================
namespace {

// std.mem.tokenizeScalar: split on ONE delimiter byte, empty tokens skipped.  Tokens are views into the text (an OBJ
// file is a few hundred thousand of them: no allocation per token).
using Token = std::string_view;
void tokenize(Token s, char delim, std::vector<Token>& out) {
  out.clear();
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && s[i] == delim) ++i;
    size_t j = i;
    while (j < s.size() && s[j] != delim) ++j;
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j;
  }
}

struct LineError {  // an error from the ObjParser.Error set or from parseFloat/parseInt
  const char* name;
};

double parseFloatToken(Token tok) {  // std.fmt.parseFloat: whole token, no whitespace
  if (tok.empty() || std::isspace(static_cast<unsigned char>(tok[0]))) throw LineError{"InvalidCharacter"};
  char small[64];
  std::string big;
  const char* text = small;
  if (tok.size() < sizeof small) {  // strtod wants a terminated string
    std::memcpy(small, tok.data(), tok.size());
    small[tok.size()] = '\0';
  } else {
    big.assign(tok);
    text = big.c_str();
  }
  errno = 0;
  char* end = nullptr;
  const double v = std::strtod(text, &end);
  if (end != text + tok.size()) throw LineError{"InvalidCharacter"};
  return v;
}

size_t parseUsizeToken(Token tok) {  // std.fmt.parseInt(usize, tok, 10)
  size_t i = 0;
  if (i < tok.size() && tok[i] == '+') ++i;
  if (i >= tok.size()) throw LineError{"InvalidCharacter"};
  size_t v = 0;
  for (; i < tok.size(); ++i) {
    const char c = tok[i];
    if (c < '0' || c > '9') throw LineError{"InvalidCharacter"};
    const size_t nv = v * 10 + static_cast<size_t>(c - '0');
    if (nv < v) throw LineError{"Overflow"};
    v = nv;
  }
  return v;
}

struct FaceVertex {
  size_t vertex_index;
  std::optional<size_t> normal_index;
};

// obj.zig:85-99 — "v", "v/t", "v/t/n", "v//n"
FaceVertex handleFaceHelper(Token token) {
  // std.mem.splitScalar keeps empty fields: field 0 is the vertex, field 2 (if there is one) the normal.
  Token parts[3];
  size_t n_parts = 0, start = 0;
  while (true) {
    const size_t p = token.find('/', start);
    const Token part = p == Token::npos ? token.substr(start) : token.substr(start, p - start);
    if (n_parts < 3) parts[n_parts] = part;
    ++n_parts;
    if (p == Token::npos) break;
    start = p + 1;
  }
  FaceVertex fv;
  fv.vertex_index = parseUsizeToken(parts[0]);
  if (n_parts < 3) return fv;  // no texture field, or no normal field
  fv.normal_index = parseUsizeToken(parts[2]);
  return fv;
}

}  // namespace

ObjParser::ObjParser() : default_group(Shape::group()) {}

void ObjParser::handleLine(std::string_view line, const InheritedState& state) {
  std::vector<Token>& tokens = line_tokens_;
  tokenize(line, ' ', tokens);
  if (tokens.empty()) throw LineError{"LineEmpty"};
  const Token first = tokens[0];
  auto tok = [&](size_t i, const char* err) -> Token {
    if (i >= tokens.size()) throw LineError{err};
    return tokens[i];
  };
  if (first == "v") {  // obj.zig:53-68
    const double x = parseFloatToken(tok(1, "IncompleteVertex"));
    const double y = parseFloatToken(tok(2, "IncompleteVertex"));
    const double z = parseFloatToken(tok(3, "IncompleteVertex"));
    // offset is a vector (w=0), so a normalised vertex carries w = 1/scale (obj.zig:66).
    vertices.push_back(Tuple::point(x, y, z).sub(offset).div(scale));
  } else if (first == "vn") {  // obj.zig:70-83
    const double x = parseFloatToken(tok(1, "IncompleteVertex"));
    const double y = parseFloatToken(tok(2, "IncompleteVertex"));
    const double z = parseFloatToken(tok(3, "IncompleteVertex"));
    normals.push_back(Tuple::vec3(x, y, z));
  } else if (first == "f") {  // obj.zig:101-150 — fan triangulation
    const FaceVertex firstv = handleFaceHelper(tok(1, "IncompleteFace"));
    FaceVertex last = handleFaceHelper(tok(2, "IncompleteFace"));
    if (tokens.size() < 4) throw LineError{"IncompleteFace"};
    auto vertexAt = [&](size_t idx) -> const Tuple& {
      if (idx == 0 || idx > vertices.size()) throw Error("IndexOutOfBounds", "face vertex " + std::to_string(idx));
      return vertices[idx - 1];  // 1-indexed
    };
    auto normalAt = [&](size_t idx) -> const Tuple& {
      if (idx == 0 || idx > normals.size()) throw Error("IndexOutOfBounds", "face normal " + std::to_string(idx));
      return normals[idx - 1];
    };
    for (size_t i = 3; i < tokens.size(); ++i) {
      const FaceVertex current = handleFaceHelper(tokens[i]);
      const Tuple& p1 = vertexAt(firstv.vertex_index);
      const Tuple& p2 = vertexAt(last.vertex_index);
      const Tuple& p3 = vertexAt(current.vertex_index);
      Shape tri = (firstv.normal_index && last.normal_index && current.normal_index)
                      ? Shape::smoothTriangle(p1, p2, p3, normalAt(*firstv.normal_index), normalAt(*last.normal_index),
                                              normalAt(*current.normal_index))
                      : Shape::triangle(p1, p2, p3);
      tri.material = state.material ? *state.material : Material{};  // copied into EVERY triangle
      tri.casts_shadow = state.casts_shadow ? *state.casts_shadow : true;
      activeGroup().addChild(std::move(tri));
      last = current;
    }
  } else if (first == "g") {  // obj.zig:152-169
    const std::string name(tok(1, "IncompleteNamedGroup"));
    default_group.addChild(Shape::group());
    active_group_ = static_cast<long>(default_group.children.size()) - 1;
    named_groups[name] = static_cast<size_t>(active_group_);
  } else {
    throw LineError{"UnknownFirstToken"};
  }
}

void ObjParser::loadObj(const std::string& obj, const InheritedState& state, bool normalize) {
  std::vector<Token> lines;
  tokenize(obj, '\n', lines);  // empty lines are skipped, never "ignored"

  if (normalize) {  // obj.zig:198-271
    double min_x = kInf, min_y = kInf, min_z = kInf;
    double max_x = -kInf, max_y = -kInf, max_z = -kInf;
    std::vector<Token> tokens;
    for (const Token line : lines) {
      tokenize(line, ' ', tokens);
      if (tokens.empty() || tokens[0] != "v") continue;
      auto coord = [&](size_t i) -> std::optional<double> {
        if (i >= tokens.size()) return std::nullopt;
        try {
          return parseFloatToken(tokens[i]);
        } catch (const LineError&) {
          return std::nullopt;
        }
      };
      if (auto x = coord(1)) {
        if (*x < min_x) min_x = *x;
        if (*x > max_x) max_x = *x;
      }
      if (auto y = coord(2)) {
        if (*y < min_y) min_y = *y;
        if (*y > max_y) max_y = *y;
      }
      if (auto z = coord(3)) {
        if (*z < min_z) min_z = *z;
        if (*z > max_z) max_z = *z;
      }
    }
    const double sx = max_x - min_x, sy = max_y - min_y, sz = max_z - min_z;
    const double x_offset = min_x + 0.5 * sx;
    const double y_offset = min_y + 0.5 * sy;
    const double z_offset = min_z + 0.5 * sz;
    const double scale_ = 0.5 * std::fmax(sx, std::fmax(sy, sz));
    offset = Tuple::vec3(x_offset, y_offset, z_offset);
    scale = scale_;
  }

  for (const Token line : lines) {
    try {
      handleLine(line, state);
    } catch (const LineError&) {
      lines_ignored += 1;  // obj.zig:277
    }
  }
}

}  // namespace rtc
