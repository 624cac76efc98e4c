// rtc_json.hpp — the small JSON reader the scene loader needs.
//
// The reference parses scene files with Zig's std.json.parseFromSlice into the
// SceneConfig struct tree (src/parsing/scene.zig:28-210) with default options:
// unknown fields, duplicate fields and missing required fields are errors.
// This reader only produces a generic value tree (object member order kept, so
// the typed layer in rtc_loader.cpp can report DuplicateField / UnknownField the
// way std.json does); numbers are converted with strtod, which — like Zig's
// parseFloat — is correctly rounded, so every f64 in a scene file gets the same
// bits as in the reference.
#pragma once
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "rtc_math.hpp"

namespace rtc::json {

struct Value {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  bool b = false;
  double num = 0;
  bool is_integer = false;  // literal had no '.', 'e' or 'E'
  std::string str;          // String payload, or the raw number literal
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  const Value* find(const std::string& key) const {
    for (const auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : s_(text) {}
  Value parseDocument() {
    Value v = parseValue();
    skipWs();
    if (pos_ != s_.size()) fail("trailing characters after JSON document");
    return v;
  }

 private:
  const std::string& s_;
  size_t pos_ = 0;
  int depth_ = 0;

  [[noreturn]] void fail(const std::string& what) const {
    throw Error("SyntaxError", what + " at byte " + std::to_string(pos_));
  }
  void skipWs() {
    while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\t' || s_[pos_] == '\n' || s_[pos_] == '\r')) ++pos_;
  }
  bool consume(char c) {
    skipWs();
    if (pos_ < s_.size() && s_[pos_] == c) {
      ++pos_;
      return true;
    }
    return false;
  }
  void expect(char c) {
    if (!consume(c)) fail(std::string("expected '") + c + "'");
  }
  Value parseValue() {
    skipWs();
    if (pos_ >= s_.size()) throw Error("UnexpectedEndOfInput");
    if (++depth_ > 256) fail("nesting too deep");
    Value v;
    const char c = s_[pos_];
    if (c == '{') {
      ++pos_;
      v.type = Value::Object;
      if (!consume('}')) {
        do {
          skipWs();
          if (pos_ >= s_.size() || s_[pos_] != '"') fail("expected object key");
          std::string key = parseString();
          expect(':');
          v.obj.emplace_back(std::move(key), parseValue());
        } while (consume(','));
        expect('}');
      }
    } else if (c == '[') {
      ++pos_;
      v.type = Value::Array;
      if (!consume(']')) {
        do {
          v.arr.push_back(parseValue());
        } while (consume(','));
        expect(']');
      }
    } else if (c == '"') {
      v.type = Value::String;
      v.str = parseString();
    } else if (s_.compare(pos_, 4, "true") == 0) {
      pos_ += 4;
      v.type = Value::Bool;
      v.b = true;
    } else if (s_.compare(pos_, 5, "false") == 0) {
      pos_ += 5;
      v.type = Value::Bool;
      v.b = false;
    } else if (s_.compare(pos_, 4, "null") == 0) {
      pos_ += 4;
      v.type = Value::Null;
    } else if (c == '-' || (c >= '0' && c <= '9')) {
      v = parseNumber();
    } else {
      fail("unexpected character");
    }
    --depth_;
    return v;
  }
  Value parseNumber() {
    const size_t start = pos_;
    bool integer = true;
    if (s_[pos_] == '-') ++pos_;
    if (pos_ >= s_.size() || s_[pos_] < '0' || s_[pos_] > '9') fail("malformed number");
    if (s_[pos_] == '0') {
      ++pos_;
    } else {
      while (pos_ < s_.size() && s_[pos_] >= '0' && s_[pos_] <= '9') ++pos_;
    }
    if (pos_ < s_.size() && s_[pos_] == '.') {
      integer = false;
      ++pos_;
      if (pos_ >= s_.size() || s_[pos_] < '0' || s_[pos_] > '9') fail("malformed number");
      while (pos_ < s_.size() && s_[pos_] >= '0' && s_[pos_] <= '9') ++pos_;
    }
    if (pos_ < s_.size() && (s_[pos_] == 'e' || s_[pos_] == 'E')) {
      integer = false;
      ++pos_;
      if (pos_ < s_.size() && (s_[pos_] == '+' || s_[pos_] == '-')) ++pos_;
      if (pos_ >= s_.size() || s_[pos_] < '0' || s_[pos_] > '9') fail("malformed number");
      while (pos_ < s_.size() && s_[pos_] >= '0' && s_[pos_] <= '9') ++pos_;
    }
    Value v;
    v.type = Value::Number;
    v.str = s_.substr(start, pos_ - start);
    v.is_integer = integer;
    v.num = std::strtod(v.str.c_str(), nullptr);
    return v;
  }
  static void appendUtf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) {
      out += static_cast<char>(cp);
    } else if (cp < 0x800) {
      out += static_cast<char>(0xC0 | (cp >> 6));
      out += static_cast<char>(0x80 | (cp & 0x3F));
    } else if (cp < 0x10000) {
      out += static_cast<char>(0xE0 | (cp >> 12));
      out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
      out += static_cast<char>(0x80 | (cp & 0x3F));
    } else {
      out += static_cast<char>(0xF0 | (cp >> 18));
      out += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
      out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
      out += static_cast<char>(0x80 | (cp & 0x3F));
    }
  }
  uint32_t parseHex4() {
    if (pos_ + 4 > s_.size()) fail("truncated \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = s_[pos_++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else fail("bad \\u escape");
    }
    return v;
  }
  std::string parseString() {
    ++pos_;  // opening quote
    std::string out;
    while (true) {
      if (pos_ >= s_.size()) throw Error("UnexpectedEndOfInput");
      const char c = s_[pos_++];
      if (c == '"') break;
      if (static_cast<unsigned char>(c) < 0x20) fail("control character in string");
      if (c != '\\') {
        out += c;
        continue;
      }
      if (pos_ >= s_.size()) throw Error("UnexpectedEndOfInput");
      const char e = s_[pos_++];
      switch (e) {
        case '"': out += '"'; break;
        case '\\': out += '\\'; break;
        case '/': out += '/'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
          uint32_t cp = parseHex4();
          if (cp >= 0xD800 && cp <= 0xDBFF && pos_ + 1 < s_.size() && s_[pos_] == '\\' && s_[pos_ + 1] == 'u') {
            pos_ += 2;
            const uint32_t lo = parseHex4();
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          appendUtf8(out, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
    return out;
  }
};

inline Value parse(const std::string& text) { return Parser(text).parseDocument(); }

}  // namespace rtc::json
