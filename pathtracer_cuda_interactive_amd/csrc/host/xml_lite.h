// Minimal XML reader for the Mitsuba-0.6 scene subset: elements, attributes,
// comments, processing instructions.  Text nodes are ignored (the format never uses them).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace pth {

struct XmlNode {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlNode>> children;

    bool has(const std::string& key) const {
        for (auto& a : attrs) if (a.first == key) return true;
        return false;
    }
    // Missing attribute -> "" (pugixml's attribute().value() contract the parser relies on)
    const std::string& attr(const std::string& key) const {
        static const std::string empty;
        for (auto& a : attrs) if (a.first == key) return a.second;
        return empty;
    }
    const XmlNode* child(const std::string& n) const {
        for (auto& c : children) if (c->name == n) return c.get();
        return nullptr;
    }
};

// Throws pth::Error(PT_ERR_PARSE) on malformed input.  Returns a synthetic root whose children are the top-level elements.
std::unique_ptr<XmlNode> xml_parse(const std::string& text);

}  // namespace pth
