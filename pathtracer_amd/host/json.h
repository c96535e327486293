// A small JSON reader for the scene configuration files (the reference uses nlohmann/json,
// src/main.cpp:15-17, which is not available as a dependency here).  Objects, arrays, strings with the
// common escapes, numbers, true / false / null.
#pragma once

#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace pt {

struct Json {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Json> arr;
  std::map<std::string, Json> obj;

  bool contains(const std::string& key) const { return kind == Object && obj.count(key) > 0; }
  const Json& operator[](const std::string& key) const
  {
    auto it = obj.find(key);
    if (kind != Object || it == obj.end()) throw std::runtime_error("json: missing key '" + key + "'");
    return it->second;
  }
  double number() const
  {
    if (kind != Number) throw std::runtime_error("json: expected a number");
    return num;
  }
  bool boolean() const
  {
    if (kind != Bool) throw std::runtime_error("json: expected a boolean");
    return b;
  }
  const std::string& string() const
  {
    if (kind != String) throw std::runtime_error("json: expected a string");
    return str;
  }
};

class JsonParser
{
 public:
  explicit JsonParser(const std::string& text) : s(text) {}
  Json parse()
  {
    Json v = value();
    ws();
    if (i != s.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& s;
  size_t i = 0;
  [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(i)); }
  void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++; }
  bool eat(char c) { ws(); if (i < s.size() && s[i] == c) { i++; return true; } return false; }
  Json value()
  {
    ws();
    if (i >= s.size()) fail("unexpected end");
    Json v;
    char c = s[i];
    if (c == '{') {
      i++;
      v.kind = Json::Object;
      if (eat('}')) return v;
      do {
        ws();
        Json k = string_value();
        if (!eat(':')) fail("expected ':'");
        v.obj[k.str] = value();
      } while (eat(','));
      if (!eat('}')) fail("expected '}'");
    } else if (c == '[') {
      i++;
      v.kind = Json::Array;
      if (eat(']')) return v;
      do v.arr.push_back(value()); while (eat(','));
      if (!eat(']')) fail("expected ']'");
    } else if (c == '"') v = string_value();
    else if (s.compare(i, 4, "true") == 0) { v.kind = Json::Bool; v.b = true; i += 4; }
    else if (s.compare(i, 5, "false") == 0) { v.kind = Json::Bool; v.b = false; i += 5; }
    else if (s.compare(i, 4, "null") == 0) { i += 4; }
    else {
      char* end = nullptr;
      v.num = std::strtod(s.c_str() + i, &end);
      if (end == s.c_str() + i) fail("unexpected character");
      v.kind = Json::Number;
      i = (size_t)(end - s.c_str());
    }
    return v;
  }
  Json string_value()
  {
    if (i >= s.size() || s[i] != '"') fail("expected a string");
    i++;
    Json v;
    v.kind = Json::String;
    while (i < s.size() && s[i] != '"') {
      char c = s[i++];
      if (c == '\\' && i < s.size()) {
        char e = s[i++];
        switch (e) {
          case 'n': c = '\n'; break;
          case 't': c = '\t'; break;
          case 'r': c = '\r'; break;
          case 'b': c = '\b'; break;
          case 'f': c = '\f'; break;
          case 'u': {  // basic multilingual plane, encoded as UTF-8
            if (i + 4 > s.size()) fail("bad \\u escape");
            unsigned cp = (unsigned)std::strtoul(s.substr(i, 4).c_str(), nullptr, 16);
            i += 4;
            if (cp < 0x80) v.str += (char)cp;
            else if (cp < 0x800) { v.str += (char)(0xC0 | (cp >> 6)); v.str += (char)(0x80 | (cp & 0x3F)); }
            else { v.str += (char)(0xE0 | (cp >> 12)); v.str += (char)(0x80 | ((cp >> 6) & 0x3F)); v.str += (char)(0x80 | (cp & 0x3F)); }
            continue;
          }
          default: c = e; break;
        }
      }
      v.str += c;
    }
    if (i >= s.size()) fail("unterminated string");
    i++;
    return v;
  }
};

}  // namespace pt
