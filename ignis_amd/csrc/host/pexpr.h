// pexpr.h — compiler from PExpr strings to the register bytecode of include/ig_expr.h.
//
// PExpr is the single-line expression language of the reference's scene files (docs/src/scene/pexpr.rst); the
// reference parses it with the third-party PExpr library (not vendored in the reference tree; CMake fetches it) and
// transpiles the typed tree to Artic (src/runtime/loader/Transpiler.cpp:960-1230). This file restates the language from
// its documentation and from the transpiler's visitor: the types (bool, int, num, vec2..4, str), the one implicit cast
// int -> num, swizzles, the operator set of the visitor callbacks (onPosNeg, onNot, onAddSub, onMulDiv, onScale, onPow,
// onMod, onAndOr, onRelOp, onEqual, onAccess), the variables of sInternalVariables (Transpiler.cpp:338-363) and the
// functions of sInternalFunctions (Transpiler.cpp:602-922) that have a counterpart in ig_expr.h. Names the table knows
// but this backend does not implement (noise, voronoi, colour-space conversions, ...) are refused by name.
#pragma once

#include "ig_expr.h"

#include <array>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace igh {
namespace pexpr {

enum class Type { Bool, Int, Num, Vec2, Vec3, Vec4, Str };

inline const char* typeName(Type t)
{
    switch (t) {
    case Type::Bool: return "bool";
    case Type::Int: return "int";
    case Type::Num: return "num";
    case Type::Vec2: return "vec2";
    case Type::Vec3: return "vec3";
    case Type::Vec4: return "vec4";
    default: return "str";
    }
}
inline bool isScalar(Type t) { return t == Type::Int || t == Type::Num; }
inline bool isVec(Type t) { return t == Type::Vec2 || t == Type::Vec3 || t == Type::Vec4; }
inline bool isArith(Type t) { return isScalar(t) || isVec(t); }
inline int lanes(Type t) { return t == Type::Vec2 ? 2 : (t == Type::Vec3 ? 3 : (t == Type::Vec4 ? 4 : 1)); }
inline Type vecOf(int n) { return n == 2 ? Type::Vec2 : (n == 3 ? Type::Vec3 : (n == 4 ? Type::Vec4 : Type::Num)); }

struct Param {
    Type type;
    std::array<float, 4> value;
};

struct Env {
    // index of the bitmap texture of that name in the texture table, -1 when the scene has no texture of that name
    std::function<int(const std::string&)> texture;
    std::map<std::string, Param> params; // the scene's "parameters" (number / vector / color)
};

struct Program {
    std::vector<uint32_t> code;
    Type type           = Type::Num;
    bool uses_view      = false; // reads V / Rd
    bool uses_frame     = false; // reads N / Nx / Ny
    bool is_const       = false;
    std::array<float, 4> value{}; // the value of a constant program
};

namespace detail {

struct Node {
    uint32_t op = IGE_CONST;
    uint32_t imm = 0;
    Type type = Type::Num;
    std::array<float, 4> cval{};
    uint32_t tex = 0;
    std::string str;
    std::vector<std::unique_ptr<Node>> args;
    bool isConst() const { return op == IGE_CONST && type != Type::Str; }
};
using NodeP = std::unique_ptr<Node>;

struct NullCtx {
    ige_v4 var(int) const { return ige_v4{}; }
    ige_v4 tex(uint32_t, float, float) const { return ige_v4{}; }
    ige_v4 evr(ige_v4, ige_v4, ige_v4 n) const { return n; }
};

[[noreturn]] inline void error(const std::string& msg) { throw std::runtime_error(msg); }

// code generation: the value of a node lands in register `base`, its arguments in base, base + 1, ...
inline void emit(const Node& n, uint32_t base, std::vector<uint32_t>& out)
{
    if (n.type == Type::Str)
        error("a string is not a value here");
    if (base + std::max<size_t>(n.args.size(), 1) > IGE_REGS)
        error("expression too deeply nested for the " + std::to_string(IGE_REGS) + " registers of the interpreter");
    for (size_t i = 0; i < n.args.size(); ++i)
        emit(*n.args[i], base + (uint32_t)i, out);
    // operand fields beyond the node's arity name `base` (a register the instruction owns), never a register past the file
    const auto r = [&](size_t i) { return i < n.args.size() ? base + (uint32_t)i : base; };
    switch (n.op) {
    case IGE_CONST:
        out.push_back(IGE_INS(IGE_CONST, base, 0, 0, 0, 0));
        for (int i = 0; i < 4; ++i) {
            uint32_t u;
            std::memcpy(&u, &n.cval[i], 4);
            out.push_back(u);
        }
        break;
    case IGE_VAR:
        out.push_back(IGE_INS(IGE_VAR, base, 0, 0, 0, n.imm));
        break;
    case IGE_TEX:
        out.push_back(IGE_INS(IGE_TEX, base, base, 0, 0, 0));
        out.push_back(n.tex);
        break;
    case IGE_BUMP:
        out.push_back(IGE_INS(IGE_BUMP, base, r(0), r(1), r(2), 0));
        out.push_back(r(3) | r(4) << 4 | r(5) << 8);
        break;
    case IGE_PACK:
        out.push_back(IGE_INS(IGE_PACK, base, r(0), r(1), r(2), r(3)));
        break;
    default:
        out.push_back(IGE_INS(n.op, base, r(0), r(1), r(2), n.imm));
        break;
    }
}

inline NodeP constant(Type t, float x, float y, float z, float w)
{
    auto n  = std::make_unique<Node>();
    n->type = t;
    n->cval = { x, y, z, w };
    return n;
}
inline NodeP scalar(Type t, float v) { return constant(t, v, v, v, v); }

// a node whose arguments are all constants is evaluated now (the reference leaves that to the Artic compiler)
inline NodeP fold(NodeP n)
{
    if (n->op == IGE_CONST || n->op == IGE_VAR || n->op == IGE_TEX || n->op == IGE_EVR)
        return n;
    for (const auto& a : n->args)
        if (!a->isConst())
            return n;
    std::vector<uint32_t> code;
    emit(*n, 0, code);
    code.push_back(IGE_INS(IGE_END, 0, 0, 0, 0, 0));
    const ige_v4 v = ige_run(code.data(), NullCtx{});
    return constant(n->type, v.v[0], v.v[1], v.v[2], v.v[3]);
}

inline NodeP make(uint32_t op, Type t, uint32_t imm, std::vector<NodeP> args)
{
    auto n  = std::make_unique<Node>();
    n->op   = op;
    n->type = t;
    n->imm  = imm;
    n->args = std::move(args);
    return fold(std::move(n));
}
template <typename... A>
inline std::vector<NodeP> list(A... a)
{
    std::vector<NodeP> v;
    (v.push_back(std::move(a)), ...);
    return v;
}

struct Token {
    enum Kind { End, Int, Num, Str, Ident, Sym } kind = End;
    std::string text;
    double num = 0;
};

class Parser {
public:
    Parser(const std::string& src, Env& env, Program& prog)
        : mSrc(src)
        , mEnv(env)
        , mProg(prog)
    {
        next();
    }

    NodeP parse()
    {
        NodeP n = parseOr();
        if (mTok.kind != Token::End)
            error("unexpected '" + mTok.text + "'");
        return n;
    }

private:
    const std::string& mSrc;
    Env& mEnv;
    Program& mProg;
    size_t mPos = 0;
    int mDepth  = 0;
    Token mTok;

    void next()
    {
        while (mPos < mSrc.size() && std::isspace((unsigned char)mSrc[mPos]))
            ++mPos;
        mTok = Token{};
        if (mPos >= mSrc.size())
            return;
        const char c = mSrc[mPos];
        if (std::isdigit((unsigned char)c) || (c == '.' && mPos + 1 < mSrc.size() && std::isdigit((unsigned char)mSrc[mPos + 1]))) {
            size_t e     = mPos;
            bool is_real = false;
            while (e < mSrc.size() && std::isdigit((unsigned char)mSrc[e]))
                ++e;
            if (e < mSrc.size() && mSrc[e] == '.' && !(e + 1 < mSrc.size() && std::isalpha((unsigned char)mSrc[e + 1]) && mSrc[e + 1] != 'e' && mSrc[e + 1] != 'E')) {
                is_real = true;
                ++e;
                while (e < mSrc.size() && std::isdigit((unsigned char)mSrc[e]))
                    ++e;
            }
            if (e < mSrc.size() && (mSrc[e] == 'e' || mSrc[e] == 'E')) {
                size_t f = e + 1;
                if (f < mSrc.size() && (mSrc[f] == '+' || mSrc[f] == '-'))
                    ++f;
                if (f < mSrc.size() && std::isdigit((unsigned char)mSrc[f])) {
                    is_real = true;
                    while (f < mSrc.size() && std::isdigit((unsigned char)mSrc[f]))
                        ++f;
                    e = f;
                }
            }
            mTok.kind = is_real ? Token::Num : Token::Int;
            mTok.text = mSrc.substr(mPos, e - mPos);
            mTok.num  = std::strtod(mTok.text.c_str(), nullptr);
            mPos      = e;
            return;
        }
        if (std::isalpha((unsigned char)c) || c == '_') {
            size_t e = mPos;
            while (e < mSrc.size() && (std::isalnum((unsigned char)mSrc[e]) || mSrc[e] == '_'))
                ++e;
            mTok.kind = Token::Ident;
            mTok.text = mSrc.substr(mPos, e - mPos);
            mPos      = e;
            return;
        }
        if (c == '"' || c == '\'') {
            const size_t e = mSrc.find(c, mPos + 1);
            if (e == std::string::npos)
                error("unterminated string");
            mTok.kind = Token::Str;
            mTok.text = mSrc.substr(mPos + 1, e - mPos - 1);
            mPos      = e + 1;
            return;
        }
        static const char* two[] = { "==", "!=", "<=", ">=", "&&", "||" };
        for (const char* t : two)
            if (mSrc.compare(mPos, 2, t) == 0) {
                mTok.kind = Token::Sym;
                mTok.text = t;
                mPos += 2;
                return;
            }
        if (std::strchr("+-*/%^(),.<>!", c)) {
            mTok.kind = Token::Sym;
            mTok.text = std::string(1, c);
            ++mPos;
            return;
        }
        error(std::string("unexpected character '") + c + "'");
    }
    bool isSym(const char* s) const { return mTok.kind == Token::Sym && mTok.text == s; }
    bool accept(const char* s)
    {
        if (!isSym(s))
            return false;
        next();
        return true;
    }
    void expect(const char* s)
    {
        if (!accept(s))
            error(std::string("expected '") + s + "' before '" + mTok.text + "'");
    }

    static void needBool(const Node& n, const char* what)
    {
        if (n.type != Type::Bool)
            error(std::string(what) + " expects bool operands, got " + typeName(n.type));
    }

    NodeP parseOr()
    {
        NodeP a = parseAnd();
        while (isSym("||")) {
            next();
            NodeP b = parseAnd();
            needBool(*a, "'||'"), needBool(*b, "'||'");
            a = make(IGE_OR, Type::Bool, 0, list(std::move(a), std::move(b)));
        }
        return a;
    }
    NodeP parseAnd()
    {
        NodeP a = parseEq();
        while (isSym("&&")) {
            next();
            NodeP b = parseEq();
            needBool(*a, "'&&'"), needBool(*b, "'&&'");
            a = make(IGE_AND, Type::Bool, 0, list(std::move(a), std::move(b)));
        }
        return a;
    }
    NodeP parseEq()
    {
        NodeP a = parseRel();
        while (isSym("==") || isSym("!=")) {
            const bool neg = mTok.text == "!=";
            next();
            NodeP b = parseRel();
            if (a->type != b->type && !(isScalar(a->type) && isScalar(b->type)))
                error(std::string("cannot compare ") + typeName(a->type) + " with " + typeName(b->type));
            if (a->type == Type::Str)
                error("strings cannot be compared");
            const uint32_t n = (uint32_t)lanes(a->type);
            a                = make(IGE_EQ, Type::Bool, n, list(std::move(a), std::move(b)));
            if (neg)
                a = make(IGE_NOT, Type::Bool, 0, list(std::move(a)));
        }
        return a;
    }
    NodeP parseRel()
    {
        NodeP a = parseAdd();
        while (isSym("<") || isSym(">") || isSym("<=") || isSym(">=")) {
            const std::string s = mTok.text;
            next();
            NodeP b = parseAdd();
            if (!isScalar(a->type) || !isScalar(b->type))
                error("'" + s + "' expects int or num operands");
            const uint32_t op = s == "<" ? IGE_LT : (s == ">" ? IGE_GT : (s == "<=" ? IGE_LE : IGE_GE));
            a                 = make(op, Type::Bool, 0, list(std::move(a), std::move(b)));
        }
        return a;
    }
    NodeP parseAdd()
    {
        NodeP a = parseMul();
        while (isSym("+") || isSym("-")) {
            const bool sub = mTok.text == "-";
            next();
            NodeP b = parseMul();
            Type t;
            if (a->type == b->type && isArith(a->type))
                t = a->type;
            else if (isScalar(a->type) && isScalar(b->type))
                t = Type::Num;
            else
                error(std::string("cannot ") + (sub ? "subtract " : "add ") + typeName(a->type) + " and " + typeName(b->type));
            a = make(sub ? IGE_SUB : IGE_ADD, t, 0, list(std::move(a), std::move(b)));
        }
        return a;
    }
    NodeP parseMul()
    {
        NodeP a = parseUnary();
        while (isSym("*") || isSym("/") || isSym("%")) {
            const char o = mTok.text[0];
            next();
            NodeP b = parseUnary();
            if (o == '%') {
                if (a->type != Type::Int || b->type != Type::Int)
                    error("'%' expects int operands");
                a = make(IGE_IMOD, Type::Int, 0, list(std::move(a), std::move(b)));
                continue;
            }
            Type t;
            if (a->type == b->type && isArith(a->type))
                t = a->type;
            else if (isScalar(a->type) && isScalar(b->type))
                t = Type::Num;
            else if (isVec(a->type) && isScalar(b->type)) // onScale: a * f, a / f
                t = a->type;
            else if (isScalar(a->type) && isVec(b->type) && o == '*') // f * a
                t = b->type;
            else
                error(std::string("cannot ") + (o == '*' ? "multiply " : "divide ") + typeName(a->type) + " and " + typeName(b->type));
            const uint32_t op = o == '*' ? IGE_MUL : (t == Type::Int ? IGE_IDIV : IGE_DIV);
            a                 = make(op, t, 0, list(std::move(a), std::move(b)));
        }
        return a;
    }
    // every cycle of the grammar (parentheses, function arguments, unary chains, '^') passes through here
    struct DepthGuard {
        int& d;
        explicit DepthGuard(int& depth)
            : d(depth)
        {
            if (++d > 128)
                error("expression too deeply nested");
        }
        ~DepthGuard() { --d; }
    };
    NodeP parseUnary()
    {
        const DepthGuard guard(mDepth);
        if (accept("+")) {
            NodeP a = parseUnary();
            if (!isArith(a->type))
                error("unary '+' expects an arithmetic operand");
            return a;
        }
        if (accept("-")) {
            NodeP a = parseUnary();
            if (!isArith(a->type))
                error("unary '-' expects an arithmetic operand");
            const Type t = a->type;
            return make(IGE_NEG, t, 0, list(std::move(a)));
        }
        if (accept("!")) {
            NodeP a = parseUnary();
            needBool(*a, "'!'");
            return make(IGE_NOT, Type::Bool, 0, list(std::move(a)));
        }
        return parsePow();
    }
    NodeP parsePow()
    {
        NodeP a = parsePostfix();
        if (accept("^")) {
            NodeP f = parseUnary();
            if (!isArith(a->type) || !isScalar(f->type))
                error("'^' expects an arithmetic base and an int or num exponent");
            // onPow: int ^ int stays int ((pow(a as f32, f as f32)) as i32), everything else is num / lane-wise
            const Type t  = a->type;
            const bool ii = t == Type::Int && f->type == Type::Int;
            a             = make(IGE_POW, (t == Type::Int && !ii) ? Type::Num : t, 0, list(std::move(a), std::move(f)));
            if (ii)
                a = make(IGE_F1, Type::Int, IGE_F_TRUNC, list(std::move(a)));
        }
        return a;
    }
    NodeP parsePostfix()
    {
        NodeP a = parsePrimary();
        while (isSym(".")) {
            next();
            if (mTok.kind != Token::Ident)
                error("expected a swizzle after '.'");
            const std::string s = mTok.text;
            next();
            if (!isVec(a->type))
                error(std::string("cannot access components of ") + typeName(a->type));
            if (s.empty() || s.size() > 4)
                error("swizzle '" + s + "' must name one to four components");
            const int in = lanes(a->type);
            uint32_t perm = 0;
            int last      = 0;
            for (size_t i = 0; i < 4; ++i) {
                int c = last;
                if (i < s.size()) {
                    static const char* names = "xyzwrgba";
                    const char* at           = std::strchr(names, s[i]);
                    if (!at || s[i] == '\0')
                        error("unknown component '" + std::string(1, s[i]) + "' in swizzle");
                    c = (int)(at - names) % 4;
                    if (c >= in)
                        error("component '" + std::string(1, s[i]) + "' is outside of " + typeName(a->type));
                }
                last = c;
                perm |= (uint32_t)c << (2 * i);
            }
            a = make(IGE_SWZ, vecOf((int)s.size()), perm, list(std::move(a)));
        }
        return a;
    }

    NodeP variable(const std::string& name)
    {
        struct V {
            const char* name;
            int id;
            Type type;
        };
        static const V vars[] = {
            { "uv", IGE_VAR_UVW, Type::Vec2 }, { "uvw", IGE_VAR_UVW, Type::Vec3 }, { "P", IGE_VAR_P, Type::Vec3 }, { "V", IGE_VAR_V, Type::Vec3 },
            { "Rd", IGE_VAR_V, Type::Vec3 }, { "N", IGE_VAR_N, Type::Vec3 }, { "Ng", IGE_VAR_NG, Type::Vec3 }, { "Nx", IGE_VAR_NX, Type::Vec3 },
            { "Ny", IGE_VAR_NY, Type::Vec3 }, { "frontside", IGE_VAR_FRONT, Type::Bool },
        };
        for (const V& v : vars)
            if (name == v.name) {
                if (v.id == IGE_VAR_V)
                    mProg.uses_view = true;
                if (v.id == IGE_VAR_N || v.id == IGE_VAR_NX || v.id == IGE_VAR_NY)
                    mProg.uses_frame = true;
                auto n  = std::make_unique<Node>();
                n->op   = IGE_VAR;
                n->imm  = (uint32_t)v.id;
                n->type = v.type;
                return n;
            }
        static const char* refused[] = { "prim_coords", "Np", "Ro", "entity_id", "Ix", "Iy", "t", "frame" };
        for (const char* r : refused)
            if (name == r)
                error("variable '" + name + "' is not supported by the HIP backend");
        if (name == "true" || name == "false")
            return scalar(Type::Bool, name == "true" ? 1.0f : 0.0f);
        if (name == "Pi")
            return scalar(Type::Num, IGM_PI);
        if (name == "E")
            return scalar(Type::Num, 2.71828182845904523536f);
        if (name == "Eps")
            return scalar(Type::Num, IGM_FLT_EPS);
        if (name == "NumMax")
            return scalar(Type::Num, IGM_FLT_MAX);
        if (name == "NumMin")
            return scalar(Type::Num, 1.17549435e-38f);
        if (name == "Inf")
            return scalar(Type::Num, INFINITY);
        if (auto it = mEnv.params.find(name); it != mEnv.params.end())
            return constant(it->second.type, it->second.value[0], it->second.value[1], it->second.value[2], it->second.value[3]);
        const int tex = mEnv.texture ? mEnv.texture(name) : -1;
        if (tex >= 0) { // a texture as a variable: looked up at uv
            auto uv  = std::make_unique<Node>();
            uv->op   = IGE_VAR;
            uv->imm  = IGE_VAR_UVW;
            uv->type = Type::Vec2;
            return texture((uint32_t)tex, std::move(uv));
        }
        error("unknown variable '" + name + "'");
    }
    static NodeP texture(uint32_t id, NodeP uv)
    {
        auto n  = std::make_unique<Node>();
        n->op   = IGE_TEX;
        n->tex  = id;
        n->type = Type::Vec4;
        n->args.push_back(std::move(uv));
        return n;
    }

    static bool allScalar(const std::vector<NodeP>& a)
    {
        for (const auto& n : a)
            if (!isScalar(n->type))
                return false;
        return !a.empty();
    }
    // the common type of arguments that must agree: equal, or int and num mixed (-> num)
    static bool commonType(const std::vector<NodeP>& a, size_t count, Type& t)
    {
        t = a[0]->type;
        for (size_t i = 1; i < count; ++i) {
            if (a[i]->type == t)
                continue;
            if (isScalar(a[i]->type) && isScalar(t))
                t = Type::Num;
            else
                return false;
        }
        return true;
    }

    NodeP call(const std::string& name, std::vector<NodeP> a)
    {
        const auto sig = [&]() {
            std::string s = name + "(";
            for (size_t i = 0; i < a.size(); ++i)
                s += std::string(i ? ", " : "") + typeName(a[i]->type);
            return s + ")";
        };
        const size_t n = a.size();
        Type t;

        static const std::map<std::string, int> f1 = {
            { "sin", IGE_F_SIN }, { "cos", IGE_F_COS }, { "tan", IGE_F_TAN }, { "asin", IGE_F_ASIN }, { "acos", IGE_F_ACOS }, { "atan", IGE_F_ATAN },
            { "exp", IGE_F_EXP }, { "exp2", IGE_F_EXP2 }, { "log", IGE_F_LOG }, { "log2", IGE_F_LOG2 }, { "log10", IGE_F_LOG10 }, { "floor", IGE_F_FLOOR },
            { "ceil", IGE_F_CEIL }, { "round", IGE_F_ROUND }, { "fract", IGE_F_FRACT }, { "trunc", IGE_F_TRUNC }, { "sqrt", IGE_F_SQRT }, { "abs", IGE_F_ABS },
            { "sign", IGE_F_SIGN }, { "rad", IGE_F_RAD }, { "deg", IGE_F_DEG }, { "smoothstep", IGE_F_SMOOTHSTEP }, { "smootherstep", IGE_F_SMOOTHERSTEP },
        };
        if (auto it = f1.find(name); it != f1.end() && n == 1 && isArith(a[0]->type)) {
            const bool keeps_int = name == "abs" || name == "sign";
            const bool scalar_only = name == "smoothstep" || name == "smootherstep";
            if (!(scalar_only && !isScalar(a[0]->type))) {
                t = (a[0]->type == Type::Int && !keeps_int) ? Type::Num : a[0]->type;
                return make(IGE_F1, t, (uint32_t)it->second, std::move(a));
            }
        }
        if (name == "int" && n == 1 && a[0]->type == Type::Num)
            return make(IGE_F1, Type::Int, IGE_F_TRUNC, std::move(a));
        if (name == "num" && n == 1 && a[0]->type == Type::Int) {
            a[0]->type = Type::Num; // ints already live as floats
            return std::move(a[0]);
        }
        if ((name == "norm" || name == "length" || name == "sum" || name == "avg") && n == 1 && isVec(a[0]->type)) {
            const uint32_t op = name == "norm" ? IGE_NORM : (name == "length" ? IGE_LENGTH : (name == "sum" ? IGE_SUM : IGE_AVG));
            const Type at     = a[0]->type;
            return make(op, name == "norm" ? at : Type::Num, (uint32_t)lanes(at), std::move(a));
        }
        if (name == "luminance" && n == 1 && a[0]->type == Type::Vec4)
            return make(IGE_LUMINANCE, Type::Num, 0, std::move(a));
        {
            // the hash noises over a vec2, with the seed or the default one (Transpiler.cpp:734-775 -> noise2_v / cellnoise2 / pnoise2 and
            // cnoise2 / ccellnoise2 / cpnoise2, *_def = DEFAULT_NOISE_SEED, src/artic/texture/noise.art:218-244)
            static const struct { const char* name; uint32_t imm; } noises[] = { { "noise", IGE_NOISE_WHITE }, { "cellnoise", IGE_NOISE_CELL }, { "pnoise", IGE_NOISE_VALUE },
                                                                                   { "cnoise", IGE_NOISE_WHITE | 4u }, { "ccellnoise", IGE_NOISE_CELL | 4u }, { "cpnoise", IGE_NOISE_VALUE | 4u },
                                                                                   { "perlin", IGE_NOISE_PERLIN }, { "sperlin", IGE_NOISE_PERLIN | 8u }, { "cperlin", IGE_NOISE_PERLIN | 4u } };
            for (const auto& f : noises) {
                if (name != f.name || (n != 1 && n != 2) || (n == 2 && a[1]->type != Type::Num && a[1]->type != Type::Int))
                    continue;
                const Type ct = a[0]->type; // a number, vec2 or vec3 of coordinates (perlin: vec2 only, there is no other form in the reference)
                const uint32_t dims = (ct == Type::Num || ct == Type::Int) ? 1u : (ct == Type::Vec2 ? 2u : (ct == Type::Vec3 ? 3u : 0u));
                if (dims == 0 || ((f.imm & 3u) == IGE_NOISE_PERLIN && dims != 2))
                    continue;
                if (n == 1)
                    a.push_back(constant(Type::Num, 36326639.0f, 36326639.0f, 36326639.0f, 36326639.0f));
                return make(IGE_NOISE, (f.imm & 4u) ? Type::Vec4 : Type::Num, f.imm | dims << 4, std::move(a));
            }
        }
        {
            // voronoi / cvoronoi / fbm / cfbm over a vec2 (Transpiler.cpp:769-792 -> voronoi2 / cvoronoi2 / fbm2 / cfbm2, src/artic/texture/voronoi.art:259-287:
            // F1, Euclidean distance; fbm: 6 octaves, lacunarity 2, gain 0.5)
            static const struct { const char* name; uint32_t imm; } cells[] = { { "voronoi", 0u }, { "cvoronoi", 4u }, { "fbm", 1u }, { "cfbm", 5u }, { "gabor", 2u } };
            for (const auto& f : cells) {
                if (name != f.name)
                    continue;
                // the reference also has forms with their parameters as arguments — voronoi(x, seed, scale, "distance", "feature"[, p]),
                // fbm(x, seed, octaves, lacunarity, gain), gabor(x, seed, impulses, ...) (Transpiler.cpp:765-795: handleVoronoiGen, fbm2_arg /
                // fbm3_arg, gabor2_gen): known, not lowered — said so instead of "no function" 
                if (n > 2)
                    error(std::string(f.name) + " with its parameters as arguments (" + std::to_string(n) + " given) is not supported by the HIP backend: only " + f.name + "(x) and " + f.name + "(x, seed) are");
                if ((n != 1 && n != 2) || (n == 2 && a[1]->type != Type::Num && a[1]->type != Type::Int))
                    continue;
                const Type ct = a[0]->type;
                const uint32_t dims = (ct == Type::Num || ct == Type::Int) ? 1u : (ct == Type::Vec2 ? 2u : (ct == Type::Vec3 ? 3u : 0u));
                // gabor: a vec2 only; fbm / cfbm: vec2 and vec3 (the reference has no fbm1: Transpiler.cpp:790-795)
                if (dims == 0 || ((f.imm & 2u) && dims != 2) || ((f.imm & 3u) == 1u && dims == 1))
                    continue;
                if (n == 1)
                    a.push_back(constant(Type::Num, 36326639.0f, 36326639.0f, 36326639.0f, 36326639.0f));
                return make(IGE_VORONOI, (f.imm & 4u) ? Type::Vec4 : Type::Num, f.imm | dims << 4, std::move(a));
            }
        }
        if (name == "hash" && n == 1 && (a[0]->type == Type::Num || a[0]->type == Type::Int)) {
            // hash_rndf(seed) (Transpiler.cpp:678, src/artic/core/random.art:91-93): the first float of the generator seeded with hash_combine(init, bits(seed)) —
            // the white noise over NO coordinate with the argument as its seed
            std::vector<NodeP> args;
            args.push_back(constant(Type::Num, 0, 0, 0, 0));
            args.push_back(std::move(a[0]));
            return make(IGE_NOISE, Type::Num, IGE_NOISE_WHITE, std::move(args));
        }
        if (name == "snoise" && (n == 1 || n == 2) && (n == 1 || a[1]->type == Type::Num || a[1]->type == Type::Int)) {
            // snoiseN(x, seed) = noiseN_v(x, seed) * 2 - 1 (src/artic/texture/noise.art:6,40,157)
            const Type ct = a[0]->type;
            const uint32_t dims = (ct == Type::Num || ct == Type::Int) ? 1u : (ct == Type::Vec2 ? 2u : (ct == Type::Vec3 ? 3u : 0u));
            if (dims) {
                if (n == 1)
                    a.push_back(constant(Type::Num, 36326639.0f, 36326639.0f, 36326639.0f, 36326639.0f));
                NodeP twice = make(IGE_MUL, Type::Num, 0, list(make(IGE_NOISE, Type::Num, IGE_NOISE_WHITE | dims << 4, std::move(a)), constant(Type::Num, 2, 2, 2, 2)));
                return make(IGE_SUB, Type::Num, 0, list(std::move(twice), constant(Type::Num, 1, 1, 1, 1)));
            }
        }
        if (name == "checkerboard" && n == 1 && (a[0]->type == Type::Vec2 || a[0]->type == Type::Vec3)) {
            const uint32_t d = (uint32_t)lanes(a[0]->type);
            return make(IGE_CHECKER, Type::Int, d, std::move(a));
        }
        if ((name == "dot" || name == "dist") && n == 2 && isVec(a[0]->type) && a[0]->type == a[1]->type) {
            const uint32_t d = (uint32_t)lanes(a[0]->type);
            return make(name == "dot" ? IGE_DOT : IGE_DIST, Type::Num, d, std::move(a));
        }
        if ((name == "cross" || name == "reflect") && n == 2 && a[0]->type == Type::Vec3 && a[1]->type == Type::Vec3)
            return make(name == "cross" ? IGE_CROSS : IGE_REFLECT, Type::Vec3, 0, std::move(a));
        if ((name == "min" || name == "max" || name == "pow" || name == "atan2" || name == "fmod") && n == 2 && commonType(a, 2, t) && isArith(t)) {
            const bool mm = name == "min" || name == "max";
            if (t == Type::Int && !mm)
                t = Type::Num;
            const uint32_t op = name == "min" ? IGE_MIN : (name == "max" ? IGE_MAX : (name == "pow" ? IGE_POW : (name == "atan2" ? IGE_ATAN2 : IGE_FMOD)));
            return make(op, t, 0, std::move(a));
        }
        if ((name == "clamp" || name == "wrap") && n == 3 && commonType(a, 3, t) && isArith(t)) {
            if (t == Type::Int && name == "wrap")
                t = Type::Num;
            return make(name == "clamp" ? IGE_CLAMP : IGE_WRAP, t, 0, std::move(a));
        }
        if (name == "mix" && n == 3 && isScalar(a[2]->type) && commonType(a, 2, t) && isArith(t))
            return make(IGE_MIX, t == Type::Int ? Type::Num : t, 0, std::move(a));
        if (name == "select" && n == 3 && a[0]->type == Type::Bool) {
            Type vt = a[1]->type;
            if (a[2]->type != vt) {
                if (isScalar(vt) && isScalar(a[2]->type))
                    vt = Type::Num;
                else
                    error("no function " + sig());
            }
            if (vt == Type::Str)
                error("select over strings is not supported by the HIP backend");
            return make(IGE_SELECT, vt, 0, std::move(a));
        }
        if ((name == "vec2" || name == "vec3" || name == "vec4" || name == "color") && allScalar(a)) {
            const int d = name == "vec2" ? 2 : (name == "vec3" ? 3 : 4);
            if (n == 1) { // vecN_expand: scalars already live in every lane
                NodeP v = std::move(a[0]);
                if (!v->isConst()) { // a copy, so that the new type does not rewrite a shared node
                    return make(IGE_SWZ, vecOf(d), 0, list(std::move(v)));
                }
                v->type = vecOf(d);
                return v;
            }
            if ((int)n == d || (name == "color" && n == 3)) {
                if (name == "color" && n == 3)
                    a.push_back(scalar(Type::Num, 1.0f)); // make_vec4(r, g, b, 1) (Transpiler.cpp:898-907)
                while (a.size() < 4)
                    a.push_back(scalar(Type::Num, 0.0f));
                return make(IGE_PACK, vecOf(d), 0, std::move(a));
            }
        }
        if (name == "bump" && n == 6 && a[0]->type == Type::Vec3 && a[1]->type == Type::Vec3 && a[2]->type == Type::Vec3 && isScalar(a[3]->type)
            && isScalar(a[4]->type) && isScalar(a[5]->type))
            return make(IGE_BUMP, Type::Vec3, 0, std::move(a));
        if (name == "ensure_valid_reflection" && n == 3 && a[0]->type == Type::Vec3 && a[1]->type == Type::Vec3 && a[2]->type == Type::Vec3)
            return make(IGE_EVR, Type::Vec3, 0, std::move(a));

        const int tex = mEnv.texture ? mEnv.texture(name) : -1;
        if (tex >= 0 && n == 1 && a[0]->type == Type::Vec2) // name(uv): the texture at other coordinates (Transpiler.cpp:1146-1149)
            return texture((uint32_t)tex, std::move(a[0]));
        if (tex >= 0 && n == 0)
            return variable(name);

        static const char* known[] = {
            "cbrt", "signbit", "rgbtoxyz", "xyztorgb", "rgbtohsv", "hsvtorgb", "rgbtohsl", "hsltorgb", "blackbody", "snap", "pingpong", "angle",
            "rotate_euler", "rotate_euler_inverse", "rotate_axis", "fresnel_dielectric", "fresnel_conductor", "noise", "snoise", "pnoise", "cellnoise",
            "perlin", "sperlin", "fbm", "voronoi", "cvoronoi", "gabor", "cnoise", "cpnoise", "ccellnoise", "cperlin", "cfbm", "smin", "smax",
            "transform_point", "transform_direction", "transform_normal", "mix_screen", "mix_overlay", "mix_dodge", "mix_burn", "mix_soft", "mix_linear",
            "mix_hue", "mix_saturation", "mix_value", "mix_color", "check_ray_flag", "lookup_curve", "curve_lookup"
        };
        for (const char* k : known)
            if (name == k)
                error("function '" + name + "' is not supported by the HIP backend");
        error("no function " + sig());
    }

    NodeP parsePrimary()
    {
        if (mTok.kind == Token::Int || mTok.kind == Token::Num) {
            NodeP n = scalar(mTok.kind == Token::Int ? Type::Int : Type::Num, (float)mTok.num);
            next();
            return n;
        }
        if (mTok.kind == Token::Str) {
            auto n  = std::make_unique<Node>();
            n->type = Type::Str;
            n->str  = mTok.text;
            next();
            return n;
        }
        if (accept("(")) {
            NodeP n = parseOr();
            expect(")");
            return n;
        }
        if (mTok.kind == Token::Ident) {
            const std::string name = mTok.text;
            next();
            if (accept("(")) {
                std::vector<NodeP> args;
                if (!accept(")")) {
                    do
                        args.push_back(parseOr());
                    while (accept(","));
                    expect(")");
                }
                return call(name, std::move(args));
            }
            return variable(name);
        }
        error(mTok.kind == Token::End ? "unexpected end of expression" : "unexpected '" + mTok.text + "'");
    }
};

} // namespace detail

// throws std::runtime_error with a message naming what is wrong or unsupported
inline Program compile(const std::string& src, Env& env)
{
    Program prog;
    detail::Parser parser(src, env, prog);
    detail::NodeP root = parser.parse();
    prog.type          = root->type;
    if (root->type == Type::Str)
        detail::error("expression is a string");
    if (root->isConst()) {
        prog.is_const = true;
        prog.value    = root->cval;
    }
    detail::emit(*root, 0, prog.code);
    prog.code.push_back(IGE_INS(IGE_END, 0, 0, 0, 0, 0));
    return prog;
}

} // namespace pexpr
} // namespace igh
