// json5.hpp — a small JSON / JSON5 reader for project and patch files.
//
// The reference reads projects with `json5::from_str` (settings/src/songs.rs:84-89) and Welsh
// patches with `serde_json` (settings/src/patches.rs:57-62).  Supported here: objects, arrays,
// strings (single or double quoted, the usual escapes), numbers (sign, fraction, exponent, hex,
// leading/trailing dot, Infinity/NaN), true / false / null, unquoted identifier keys, trailing
// commas, // and /* */ comments.
#pragma once
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace json5 {

struct Value;
using ValuePtr = std::shared_ptr<Value>;
struct Value {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<ValuePtr> arr;
  std::vector<std::pair<std::string, ValuePtr>> obj; // insertion order kept (serde enums are single-key maps)

  bool is_null() const { return type == Null; }
  bool is_string() const { return type == String; }
  bool is_object() const { return type == Object; }
  bool is_array() const { return type == Array; }
  bool is_number() const { return type == Number; }
  const Value* get(const std::string& key) const {
    if (type != Object) return nullptr;
    for (auto& kv : obj) if (kv.first == key) return kv.second.get();
    return nullptr;
  }
  double number_or(const std::string& key, double dflt) const {
    const Value* v = get(key);
    return v && v->type == Number ? v->num : dflt;
  }
  std::string string_or(const std::string& key, const std::string& dflt) const {
    const Value* v = get(key);
    return v && v->type == String ? v->str : dflt;
  }
  bool bool_or(const std::string& key, bool dflt) const {
    const Value* v = get(key);
    return v && v->type == Bool ? v->b : dflt;
  }
};

// double -> integer for numbers that come from a file: NaN gives `dflt`, everything else is clamped
// (a double outside the integer's range is undefined behaviour to cast).
inline long long to_int(double x, long long lo, long long hi, long long dflt = 0) {
  if (!(x == x)) return dflt;
  if (x <= (double)lo) return lo;
  if (x >= (double)hi) return hi;
  return (long long)x;
}

class Parser {
 public:
  explicit Parser(const std::string& text) : s_(text) {}
  ValuePtr parse() {
    ValuePtr v = value();
    ws();
    if (i_ != s_.size()) fail("trailing characters");
    return v;
  }

 private:
  [[noreturn]] void fail(const std::string& what) const {
    size_t line = 1, col = 1;
    for (size_t k = 0; k < i_ && k < s_.size(); ++k) { if (s_[k] == '\n') { ++line; col = 1; } else ++col; }
    throw std::runtime_error("JSON5 parse error at line " + std::to_string(line) + " column " + std::to_string(col) + ": " + what);
  }
  void ws() {
    for (;;) {
      while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) ++i_;
      if (i_ + 1 < s_.size() && s_[i_] == '/' && s_[i_ + 1] == '/') { while (i_ < s_.size() && s_[i_] != '\n') ++i_; continue; }
      if (i_ + 1 < s_.size() && s_[i_] == '/' && s_[i_ + 1] == '*') {
        i_ += 2;
        while (i_ + 1 < s_.size() && !(s_[i_] == '*' && s_[i_ + 1] == '/')) ++i_;
        if (i_ + 1 >= s_.size()) fail("unterminated comment");
        i_ += 2;
        continue;
      }
      return;
    }
  }
  static bool ident_start(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_' || c == '$'; }
  static bool ident_char(char c) { return ident_start(c) || (c >= '0' && c <= '9') || c == '-'; }
  std::string quoted() {
    const char q = s_[i_++];
    std::string out;
    while (i_ < s_.size() && s_[i_] != q) {
      char c = s_[i_++];
      if (c == '\\') {
        if (i_ >= s_.size()) fail("bad escape");
        char e = s_[i_++];
        switch (e) {
          case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
          case 'b': out += '\b'; break; case 'f': out += '\f'; break; case '0': out += '\0'; break;
          case '\n': break; // line continuation
          case 'u': {
            if (i_ + 4 > s_.size()) fail("bad \\u escape");
            unsigned cp = (unsigned)std::strtoul(s_.substr(i_, 4).c_str(), nullptr, 16);
            i_ += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += e; break;
        }
      } else out += c;
    }
    if (i_ >= s_.size()) fail("unterminated string");
    ++i_;
    return out;
  }
  struct DepthGuard {
    size_t& d;
    explicit DepthGuard(size_t& depth) : d(depth) { ++d; }
    ~DepthGuard() { --d; }
  };
  ValuePtr value() {
    ws();
    if (i_ >= s_.size()) fail("unexpected end of input");
    DepthGuard guard(depth_);
    if (depth_ > kMaxDepth) fail("nesting deeper than 256 levels"); // value() recurses: bound the stack for hostile input
    auto v = std::make_shared<Value>();
    const char c = s_[i_];
    if (c == '{') {
      v->type = Value::Object;
      ++i_;
      for (;;) {
        ws();
        if (i_ < s_.size() && s_[i_] == '}') { ++i_; break; }
        std::string key;
        if (i_ < s_.size() && (s_[i_] == '"' || s_[i_] == '\'')) key = quoted();
        else if (i_ < s_.size() && ident_start(s_[i_])) { while (i_ < s_.size() && ident_char(s_[i_])) key += s_[i_++]; }
        else fail("expected a key");
        ws();
        if (i_ >= s_.size() || s_[i_] != ':') fail("expected ':'");
        ++i_;
        v->obj.emplace_back(key, value());
        ws();
        if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
        if (i_ < s_.size() && s_[i_] == '}') { ++i_; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v->type = Value::Array;
      ++i_;
      for (;;) {
        ws();
        if (i_ < s_.size() && s_[i_] == ']') { ++i_; break; }
        v->arr.push_back(value());
        ws();
        if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
        if (i_ < s_.size() && s_[i_] == ']') { ++i_; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"' || c == '\'') {
      v->type = Value::String;
      v->str = quoted();
    } else if (s_.compare(i_, 4, "true") == 0) { v->type = Value::Bool; v->b = true; i_ += 4; }
    else if (s_.compare(i_, 5, "false") == 0) { v->type = Value::Bool; v->b = false; i_ += 5; }
    else if (s_.compare(i_, 4, "null") == 0) { v->type = Value::Null; i_ += 4; }
    else {
      v->type = Value::Number;
      size_t j = i_;
      double sign = 1.0;
      if (s_[j] == '+' || s_[j] == '-') { if (s_[j] == '-') sign = -1.0; ++j; }
      if (s_.compare(j, 8, "Infinity") == 0) { v->num = sign * INFINITY; i_ = j + 8; }
      else if (s_.compare(j, 3, "NaN") == 0) { v->num = NAN; i_ = j + 3; }
      else if (j + 1 < s_.size() && s_[j] == '0' && (s_[j + 1] == 'x' || s_[j + 1] == 'X')) {
        char* end = nullptr;
        v->num = sign * (double)std::strtoull(s_.c_str() + j + 2, &end, 16);
        if (end == s_.c_str() + j + 2) fail("bad hex number");
        i_ = (size_t)(end - s_.c_str());
      } else {
        char* end = nullptr;
        v->num = std::strtod(s_.c_str() + i_, &end);
        if (end == s_.c_str() + i_) fail("unexpected character");
        i_ = (size_t)(end - s_.c_str());
      }
    }
    return v;
  }
  const std::string& s_;
  size_t i_ = 0;
  size_t depth_ = 0;
  static constexpr size_t kMaxDepth = 256;
};

inline ValuePtr parse(const std::string& text) { return Parser(text).parse(); }

} // namespace json5
