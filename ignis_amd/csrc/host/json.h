// Minimal JSON reader for the Ignis scene format (the reference uses RapidJSON,
// src/runtime/loader/Parser.cpp). Object member order is preserved because this
// backend assigns ids in declaration order.
#pragma once

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace igh {

struct JsonValue {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b       = false;
    double num   = 0;
    bool is_int  = false;
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;

    bool isNull() const { return type == Null; }
    bool isBool() const { return type == Bool; }
    bool isNumber() const { return type == Number; }
    bool isString() const { return type == String; }
    bool isArray() const { return type == Array; }
    bool isObject() const { return type == Object; }

    const JsonValue* find(const std::string& key) const
    {
        if (type != Object)
            return nullptr;
        for (const auto& p : obj)
            if (p.first == key)
                return &p.second;
        return nullptr;
    }
    bool has(const std::string& key) const { return find(key) != nullptr; }

    float getNumber(const std::string& key, float def) const
    {
        const JsonValue* v = find(key);
        return (v && v->isNumber()) ? (float)v->num : def;
    }
    int getInt(const std::string& key, int def) const
    {
        const JsonValue* v = find(key);
        return (v && v->isNumber()) ? (int)v->num : def;
    }
    bool getBool(const std::string& key, bool def) const
    {
        const JsonValue* v = find(key);
        return (v && v->isBool()) ? v->b : def;
    }
    std::string getString(const std::string& key, const std::string& def = "") const
    {
        const JsonValue* v = find(key);
        return (v && v->isString()) ? v->str : def;
    }
};

class JsonParser {
public:
    explicit JsonParser(const std::string& text)
        : s(text)
    {
    }

    JsonValue parse()
    {
        skip();
        JsonValue v = value(0);
        skip();
        if (pos != s.size())
            fail("trailing characters");
        return v;
    }

private:
    const std::string& s;
    size_t pos = 0;

    [[noreturn]] void fail(const std::string& msg) const
    {
        throw std::runtime_error("JSON error at offset " + std::to_string(pos) + ": " + msg);
    }

    void skip()
    {
        for (;;) {
            while (pos < s.size() && std::isspace((unsigned char)s[pos]))
                ++pos;
            // Tolerate // and /* */ comments like RapidJSON's kParseCommentsFlag
            if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '/') {
                while (pos < s.size() && s[pos] != '\n')
                    ++pos;
            } else if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '*') {
                pos += 2;
                while (pos + 1 < s.size() && !(s[pos] == '*' && s[pos + 1] == '/'))
                    ++pos;
                if (pos + 1 >= s.size())
                    fail("unterminated comment");
                pos += 2;
            } else {
                return;
            }
        }
    }

    JsonValue value(int depth)
    {
        if (depth > 256)
            fail("nesting too deep");
        if (pos >= s.size())
            fail("unexpected end");
        const char c = s[pos];
        if (c == '{')
            return object(depth);
        if (c == '[')
            return array(depth);
        if (c == '"') {
            JsonValue v;
            v.type = JsonValue::String;
            v.str  = string();
            return v;
        }
        if (c == 't' || c == 'f' || c == 'n')
            return literal();
        if (c == '-' || c == '+' || std::isdigit((unsigned char)c))
            return number();
        fail(std::string("unexpected character '") + c + "'");
    }

    JsonValue literal()
    {
        JsonValue v;
        if (s.compare(pos, 4, "true") == 0) {
            v.type = JsonValue::Bool;
            v.b    = true;
            pos += 4;
        } else if (s.compare(pos, 5, "false") == 0) {
            v.type = JsonValue::Bool;
            v.b    = false;
            pos += 5;
        } else if (s.compare(pos, 4, "null") == 0) {
            pos += 4;
        } else {
            fail("bad literal");
        }
        return v;
    }

    JsonValue number()
    {
        const char* begin = s.c_str() + pos;
        char* end         = nullptr;
        const double d    = std::strtod(begin, &end);
        if (end == begin)
            fail("bad number");
        JsonValue v;
        v.type   = JsonValue::Number;
        v.num    = d;
        v.is_int = true;
        for (const char* p = begin; p != end; ++p)
            if (*p == '.' || *p == 'e' || *p == 'E')
                v.is_int = false;
        pos += (size_t)(end - begin);
        return v;
    }

    std::string string()
    {
        ++pos; // opening quote
        std::string out;
        while (pos < s.size() && s[pos] != '"') {
            char c = s[pos++];
            if (c == '\\') {
                if (pos >= s.size())
                    fail("bad escape");
                char e = s[pos++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (pos + 4 > s.size())
                        fail("bad unicode escape");
                    unsigned cp = (unsigned)std::strtoul(s.substr(pos, 4).c_str(), nullptr, 16);
                    pos += 4;
                    if (cp < 0x80) {
                        out += (char)cp;
                    } else if (cp < 0x800) {
                        out += (char)(0xC0 | (cp >> 6));
                        out += (char)(0x80 | (cp & 0x3F));
                    } else {
                        out += (char)(0xE0 | (cp >> 12));
                        out += (char)(0x80 | ((cp >> 6) & 0x3F));
                        out += (char)(0x80 | (cp & 0x3F));
                    }
                } break;
                default: out += e; break;
                }
            } else {
                out += c;
            }
        }
        if (pos >= s.size())
            fail("unterminated string");
        ++pos; // closing quote
        return out;
    }

    JsonValue array(int depth)
    {
        JsonValue v;
        v.type = JsonValue::Array;
        ++pos;
        skip();
        if (pos < s.size() && s[pos] == ']') {
            ++pos;
            return v;
        }
        for (;;) {
            skip();
            v.arr.push_back(value(depth + 1));
            skip();
            if (pos >= s.size())
                fail("unterminated array");
            if (s[pos] == ',') {
                ++pos;
                skip();
                if (pos < s.size() && s[pos] == ']') { // trailing comma
                    ++pos;
                    return v;
                }
                continue;
            }
            if (s[pos] == ']') {
                ++pos;
                return v;
            }
            fail("expected ',' or ']'");
        }
    }

    JsonValue object(int depth)
    {
        JsonValue v;
        v.type = JsonValue::Object;
        ++pos;
        skip();
        if (pos < s.size() && s[pos] == '}') {
            ++pos;
            return v;
        }
        for (;;) {
            skip();
            if (pos >= s.size() || s[pos] != '"')
                fail("expected member name");
            std::string key = string();
            skip();
            if (pos >= s.size() || s[pos] != ':')
                fail("expected ':'");
            ++pos;
            skip();
            v.obj.emplace_back(std::move(key), value(depth + 1));
            skip();
            if (pos >= s.size())
                fail("unterminated object");
            if (s[pos] == ',') {
                ++pos;
                skip();
                if (pos < s.size() && s[pos] == '}') {
                    ++pos;
                    return v;
                }
                continue;
            }
            if (s[pos] == '}') {
                ++pos;
                return v;
            }
            fail("expected ',' or '}'");
        }
    }
};

} // namespace igh
