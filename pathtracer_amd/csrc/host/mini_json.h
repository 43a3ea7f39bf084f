// mini_json.h — small recursive-descent JSON reader, enough for glTF 2.0 documents.
// Replaces the role tinygltf + nlohmann json play in the reference loader
// (src/scene.cu:14-17); only the value model the glTF reader needs is kept.
#pragma once

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace pth {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    // insertion order kept separately: glTF attribute maps are iterated in
    // key order by tinygltf (std::map), so an ordered map is what we want.
    std::map<std::string, JsonValue> obj;

    bool is_null() const { return kind == Null; }
    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool is_number() const { return kind == Number; }
    bool is_string() const { return kind == String; }

    const JsonValue *find(const std::string &key) const {
        if (kind != Object) return nullptr;
        auto it = obj.find(key);
        return it == obj.end() ? nullptr : &it->second;
    }
    bool has(const std::string &key) const { return find(key) != nullptr; }
    const JsonValue &at(const std::string &key) const {
        static const JsonValue null_value;
        const JsonValue *v = find(key);
        return v ? *v : null_value;
    }
    double number_or(double dflt) const { return kind == Number ? num : dflt; }
    int int_or(int dflt) const { return kind == Number ? (int)num : dflt; }
    std::string string_or(const std::string &dflt) const { return kind == String ? str : dflt; }
    size_t size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
    const JsonValue &operator[](size_t i) const { return arr[i]; }
};

class JsonParser {
  public:
    explicit JsonParser(const std::string &text) : s_(text), p_(0) {}

    JsonValue parse() {
        JsonValue v = value();
        ws();
        if (p_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s_;
    size_t p_;

    [[noreturn]] void fail(const char *what) const {
        throw std::runtime_error(std::string("json: ") + what + " at byte " + std::to_string(p_));
    }
    void ws() {
        while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\t' || s_[p_] == '\n' || s_[p_] == '\r')) p_++;
    }
    char peek() {
        ws();
        if (p_ >= s_.size()) fail("unexpected end");
        return s_[p_];
    }
    void expect(char c) {
        if (peek() != c) fail("unexpected character");
        p_++;
    }
    bool lit(const char *word) {
        size_t n = strlen(word);
        if (s_.compare(p_, n, word) == 0) {
            p_ += n;
            return true;
        }
        return false;
    }

    JsonValue value() {
        char c = peek();
        JsonValue v;
        if (c == '{') {
            v.kind = JsonValue::Object;
            p_++;
            if (peek() == '}') {
                p_++;
                return v;
            }
            for (;;) {
                if (peek() != '"') fail("object key");
                std::string key = string();
                expect(':');
                v.obj[key] = value();
                char d = peek();
                p_++;
                if (d == '}') break;
                if (d != ',') fail("object separator");
            }
        } else if (c == '[') {
            v.kind = JsonValue::Array;
            p_++;
            if (peek() == ']') {
                p_++;
                return v;
            }
            for (;;) {
                v.arr.push_back(value());
                char d = peek();
                p_++;
                if (d == ']') break;
                if (d != ',') fail("array separator");
            }
        } else if (c == '"') {
            v.kind = JsonValue::String;
            v.str = string();
        } else if (lit("true")) {
            v.kind = JsonValue::Bool;
            v.b = true;
        } else if (lit("false")) {
            v.kind = JsonValue::Bool;
        } else if (lit("null")) {
            v.kind = JsonValue::Null;
        } else {
            const char *start = s_.c_str() + p_;
            char *end = nullptr;
            double d = strtod(start, &end);
            if (end == start) fail("number");
            p_ += (size_t)(end - start);
            v.kind = JsonValue::Number;
            v.num = d;
        }
        return v;
    }

    static void append_utf8(std::string &out, unsigned cp) {
        if (cp < 0x80) {
            out.push_back((char)cp);
        } else if (cp < 0x800) {
            out.push_back((char)(0xC0 | (cp >> 6)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18)));
            out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }

    unsigned hex4() {
        if (p_ + 4 > s_.size()) fail("\\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; i++) {
            char c = s_[p_++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else fail("\\u escape");
        }
        return v;
    }

    std::string string() {
        expect('"');
        std::string out;
        for (;;) {
            if (p_ >= s_.size()) fail("unterminated string");
            char c = s_[p_++];
            if (c == '"') break;
            if (c != '\\') {
                out.push_back(c);
                continue;
            }
            if (p_ >= s_.size()) fail("escape");
            char e = s_[p_++];
            switch (e) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                unsigned cp = hex4();
                if (cp >= 0xD800 && cp < 0xDC00 && s_.compare(p_, 2, "\\u") == 0) {
                    p_ += 2;
                    unsigned lo = hex4();
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                append_utf8(out, cp);
                break;
            }
            default: fail("escape");
            }
        }
        return out;
    }
};

} // namespace pth
