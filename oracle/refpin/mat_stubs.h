// ORACLE pin (test infrastructure only): what Rtxpt/Materials/MaterialsBaker.{h,cpp} (PTMaterial defaults, Read, FillData, GetBindlessTextureIndex,
// IsEmissive) need from outside the reference tree in order to compile as they stand: a jsoncpp-style Json::Value with Donut's `>>` extraction
// (donut/core/json.h: a field that is missing or of the wrong kind leaves the destination untouched), Donut's LoadedTexture / TextureCache, dm::float3,
// the log. Generated translation unit: oracle/refpin/hlsl_tu.py --materials; never written to disk.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>
#include <cassert>
#include <algorithm>
#include <filesystem>
#include <map>
#include <optional>
#include <memory>
#include <string>
#include <vector>
#include "hlsl_shim.h"

namespace Json {
struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false; double num = 0; std::string str; std::vector<Value> arr; std::map<std::string, Value> obj;
    bool isArray() const { return kind == Array; } size_t size() const { return arr.size(); } std::string asString() const { return str; }
    std::vector<Value>::const_iterator begin() const { return arr.begin(); } std::vector<Value>::const_iterator end() const { return arr.end(); }
    bool empty() const { return kind == Null || (kind == Array && arr.empty()) || (kind == Object && obj.empty()); }
    Value operator[](const std::string& k) const { if (kind != Object) return Value(); auto it = obj.find(k); return it == obj.end() ? Value() : it->second; }
};
// minimal recursive-descent parser (test documents only)
struct Parser { const char* p; bool ok = true;
    void ws() { while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') p++; }
    Value value() { ws(); Value v;
        if (*p == '{') { p++; v.kind = Value::Object; ws(); if (*p == '}') { p++; return v; }
            for (;;) { ws(); Value k = value(); ws(); if (k.kind != Value::String || *p != ':') { ok = false; return v; } p++; v.obj[k.str] = value(); ws(); if (*p == ',') { p++; continue; } if (*p == '}') { p++; break; } ok = false; return v; } }
        else if (*p == '[') { p++; v.kind = Value::Array; ws(); if (*p == ']') { p++; return v; }
            for (;;) { v.arr.push_back(value()); ws(); if (*p == ',') { p++; continue; } if (*p == ']') { p++; break; } ok = false; return v; } }
        else if (*p == '"') { p++; v.kind = Value::String; while (*p && *p != '"') { if (*p == '\\' && p[1]) p++; v.str.push_back(*p++); } if (*p == '"') p++; else ok = false; }
        else if (!strncmp(p, "true", 4)) { p += 4; v.kind = Value::Bool; v.b = true; } else if (!strncmp(p, "false", 5)) { p += 5; v.kind = Value::Bool; }
        else if (!strncmp(p, "null", 4)) { p += 4; }
        else { char* e; v.num = strtod(p, &e); if (e == p) ok = false; else { p = e; v.kind = Value::Number; } }
        return v; } };
}
namespace dm { using hl::float3; using hl::uint; }
typedef hl::uint uint;
namespace donut { namespace math { using namespace hl; static inline float log2f(float v) { return hl::log2(v); } static inline bool any(hl::bool3 b) { return hl::any(b); } }
namespace log { template <class... A> void warning(A...) {} }
namespace engine {
    struct TextureDesc { uint width = 1, height = 1, mipLevels = 1; };
    struct TextureHandleStub { TextureDesc d; TextureDesc getDesc() const { return d; } };
    struct DescriptorStub { uint v = ~0u; uint Get() const { return v; } };
    struct LoadedTexture { std::shared_ptr<TextureHandleStub> texture; DescriptorStub bindlessDescriptor; std::string path; };
    struct Material {};
    // the pin's "texture cache": the caller registers, per document, the packed texture word its loader produced for each path (0xFFFFFFFF: not loadable)
    struct TextureCache { std::map<std::string, uint> words;
        std::shared_ptr<LoadedTexture> LoadTextureFromFileDeferred(const std::filesystem::path& p, bool) {
            auto it = words.find(p.filename().string());
            if (it == words.end() || it->second == 0xFFFFFFFFu) return nullptr;
            auto t = std::make_shared<LoadedTexture>(); t->texture = std::make_shared<TextureHandleStub>();
            uint w = it->second; t->bindlessDescriptor.v = w & 0xFFFFu; t->texture->d.mipLevels = (w >> 16) & 0xFFu; t->texture->d.width = 1u << (w >> 24); t->texture->d.height = 1;     // log2(w*h) = baseLOD
            t->path = p.string(); return t; } };
} }
using donut::engine::LoadedTexture;
// donut/core/json.h extraction: only a value of the matching kind is taken
static inline void operator>>(const Json::Value& n, std::string& d) { if (n.kind == Json::Value::String) d = n.str; }
static inline void operator>>(const Json::Value& n, bool& d) { if (n.kind == Json::Value::Bool) d = n.b; }
static inline void operator>>(const Json::Value& n, float& d) { if (n.kind == Json::Value::Number) d = (float)n.num; }
static inline void operator>>(const Json::Value& n, int& d) { if (n.kind == Json::Value::Number) d = (int)n.num; }
static inline void operator>>(const Json::Value& n, uint& d) { if (n.kind == Json::Value::Number) d = (uint)n.num; }
static inline void operator>>(const Json::Value& n, dm::float3& d) { if (n.kind == Json::Value::Array && n.arr.size() == 3 && n.arr[0].kind == Json::Value::Number) d = dm::float3((float)n.arr[0].num, (float)n.arr[1].num, (float)n.arr[2].num); }
using std::min; using std::max; using std::clamp;
struct MaterialShaderPermutation {};
struct PTMaterialBase { virtual ~PTMaterialBase() {} std::string Name, ModelName;
    virtual void Write(Json::Value&) = 0; virtual bool Read(Json::Value&, const std::filesystem::path&, const std::shared_ptr<donut::engine::TextureCache>&) = 0;
    virtual bool HasAlphaTest() const = 0; virtual MaterialShaderPermutation ComputeShaderPermutation(const std::string&) = 0; };
