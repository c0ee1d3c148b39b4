#include "xml_lite.h"

#include <cctype>

#include "parsed_scene.h"

namespace pth {
namespace {

struct Cursor {
    const std::string& s;
    size_t i = 0;
    explicit Cursor(const std::string& t) : s(t) {}
    bool eof() const { return i >= s.size(); }
    bool starts(const char* lit) const { return s.compare(i, std::char_traits<char>::length(lit), lit) == 0; }
    void skip_ws() { while (!eof() && std::isspace((unsigned char)s[i])) i++; }
    [[noreturn]] void fail(const std::string& why) const {
        throw Error(PT_ERR_PARSE, "XML parse error at offset " + std::to_string(i) + ": " + why);
    }
    void skip_until(const char* lit) {
        size_t p = s.find(lit, i);
        if (p == std::string::npos) fail(std::string("unterminated construct, expected ") + lit);
        i = p + std::char_traits<char>::length(lit);
    }
};

std::string decode_entities(const std::string& v) {
    if (v.find('&') == std::string::npos) return v;
    static const std::pair<const char*, char> ents[] = {{"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
    std::string out;
    for (size_t i = 0; i < v.size();) {
        bool done = false;
        if (v[i] == '&')
            for (auto& e : ents) {
                size_t n = std::char_traits<char>::length(e.first);
                if (v.compare(i, n, e.first) == 0) { out.push_back(e.second); i += n; done = true; break; }
            }
        if (!done) out.push_back(v[i++]);
    }
    return out;
}

std::string read_name(Cursor& c) {
    size_t b = c.i;
    while (!c.eof() && (std::isalnum((unsigned char)c.s[c.i]) || c.s[c.i] == '_' || c.s[c.i] == '-' || c.s[c.i] == ':' || c.s[c.i] == '.')) c.i++;
    if (c.i == b) c.fail("expected a name");
    return c.s.substr(b, c.i - b);
}

// Skips comments / PIs / doctype / text; returns true when positioned at the '<' of an element open or close tag.
bool skip_misc(Cursor& c) {
    for (;;) {
        while (!c.eof() && c.s[c.i] != '<') c.i++;       // text content is ignored
        if (c.eof()) return false;
        if (c.starts("<!--")) { c.skip_until("-->"); continue; }
        if (c.starts("<?")) { c.skip_until("?>"); continue; }
        if (c.starts("<!")) { c.skip_until(">"); continue; }
        return true;
    }
}

std::unique_ptr<XmlNode> parse_element(Cursor& c, int depth) {
    if (depth > 64) c.fail("nesting too deep");
    c.i++;  // '<'
    auto node = std::make_unique<XmlNode>();
    node->name = read_name(c);
    for (;;) {
        c.skip_ws();
        if (c.eof()) c.fail("unterminated tag");
        if (c.starts("/>")) { c.i += 2; return node; }
        if (c.s[c.i] == '>') { c.i++; break; }
        std::string key = read_name(c);
        c.skip_ws();
        if (c.eof() || c.s[c.i] != '=') c.fail("expected '=' after attribute name");
        c.i++;
        c.skip_ws();
        if (c.eof() || (c.s[c.i] != '"' && c.s[c.i] != '\'')) c.fail("expected quoted attribute value");
        char q = c.s[c.i++];
        size_t e = c.s.find(q, c.i);
        if (e == std::string::npos) c.fail("unterminated attribute value");
        node->attrs.emplace_back(key, decode_entities(c.s.substr(c.i, e - c.i)));
        c.i = e + 1;
    }
    for (;;) {
        if (!skip_misc(c)) c.fail("missing </" + node->name + ">");
        if (c.starts("</")) {
            c.i += 2;
            std::string n = read_name(c);
            if (n != node->name) c.fail("mismatched closing tag </" + n + "> for <" + node->name + ">");
            c.skip_ws();
            if (c.eof() || c.s[c.i] != '>') c.fail("expected '>'");
            c.i++;
            return node;
        }
        node->children.push_back(parse_element(c, depth + 1));
    }
}

}  // namespace

std::unique_ptr<XmlNode> xml_parse(const std::string& text) {
    Cursor c(text);
    auto root = std::make_unique<XmlNode>();
    while (skip_misc(c)) {
        if (c.starts("</")) c.fail("unexpected closing tag");
        root->children.push_back(parse_element(c, 0));
    }
    return root;
}

}  // namespace pth
