// xml_mini.hpp — a minimal XML reader sufficient for MJCF files: elements, attributes,
// comments, declarations, self-closing tags.  Text nodes are ignored (MJCF carries no text).
#pragma once
#include <cctype>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace hb {

struct XmlNode {
  std::string name;
  std::vector<std::pair<std::string, std::string>> attrs;  // document order
  std::vector<std::unique_ptr<XmlNode>> children;
  const std::string* attr(const std::string& k) const {
    for (auto& a : attrs) if (a.first == k) return &a.second;
    return nullptr;
  }
  bool has(const std::string& k) const { return attr(k) != nullptr; }
  const XmlNode* child(const std::string& n) const {
    for (auto& c : children) if (c->name == n) return c.get();
    return nullptr;
  }
};

class XmlParser {
 public:
  explicit XmlParser(const std::string& s) : s_(s), p_(0) {}
  std::unique_ptr<XmlNode> parse(std::string& err) {
    skip_misc();
    auto root = element(err);
    if (!root && err.empty()) err = "xml: no root element";
    return root;
  }

 private:
  const std::string& s_;
  size_t p_;
  bool starts(const char* t) const { return s_.compare(p_, strlen(t), t) == 0; }
  void skip_ws() { while (p_ < s_.size() && isspace((unsigned char)s_[p_])) p_++; }
  void skip_misc() {  // whitespace, comments, declarations, doctype, stray text
    for (;;) {
      skip_ws();
      if (p_ >= s_.size()) return;
      if (starts("<!--")) { size_t e = s_.find("-->", p_ + 4); p_ = (e == std::string::npos) ? s_.size() : e + 3; }
      else if (starts("<?")) { size_t e = s_.find("?>", p_ + 2); p_ = (e == std::string::npos) ? s_.size() : e + 2; }
      else if (starts("<!")) { size_t e = s_.find('>', p_ + 2); p_ = (e == std::string::npos) ? s_.size() : e + 1; }
      else if (s_[p_] != '<') { size_t e = s_.find('<', p_); p_ = (e == std::string::npos) ? s_.size() : e; }
      else return;
    }
  }
  static bool name_char(char c) { return isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }
  std::string name() { size_t b = p_; while (p_ < s_.size() && name_char(s_[p_])) p_++; return s_.substr(b, p_ - b); }
  std::string line_of(size_t pos) const {
    size_t ln = 1;
    for (size_t i = 0; i < pos && i < s_.size(); i++) if (s_[i] == '\n') ln++;
    return std::to_string(ln);
  }
  std::unique_ptr<XmlNode> element(std::string& err) {
    if (p_ >= s_.size() || s_[p_] != '<') return nullptr;
    p_++;
    std::unique_ptr<XmlNode> n(new XmlNode);
    n->name = name();
    if (n->name.empty()) { err = "xml: bad tag at line " + line_of(p_); return nullptr; }
    for (;;) {
      skip_ws();
      if (p_ >= s_.size()) { err = "xml: unexpected end in <" + n->name + ">"; return nullptr; }
      if (starts("/>")) { p_ += 2; return n; }
      if (s_[p_] == '>') { p_++; break; }
      std::string k = name();
      skip_ws();
      if (k.empty() || p_ >= s_.size() || s_[p_] != '=') { err = "xml: bad attribute in <" + n->name + "> at line " + line_of(p_); return nullptr; }
      p_++;
      skip_ws();
      char q = p_ < s_.size() ? s_[p_] : 0;
      if (q != '"' && q != '\'') { err = "xml: unquoted attribute in <" + n->name + "> at line " + line_of(p_); return nullptr; }
      size_t e = s_.find(q, p_ + 1);
      if (e == std::string::npos) { err = "xml: unterminated attribute at line " + line_of(p_); return nullptr; }
      n->attrs.emplace_back(k, s_.substr(p_ + 1, e - p_ - 1));
      p_ = e + 1;
    }
    for (;;) {  // children until matching close tag
      skip_misc();
      if (p_ >= s_.size()) { err = "xml: missing </" + n->name + ">"; return nullptr; }
      if (starts("</")) {
        p_ += 2;
        std::string c = name();
        skip_ws();
        if (c != n->name || p_ >= s_.size() || s_[p_] != '>') { err = "xml: mismatched </" + c + "> for <" + n->name + "> at line " + line_of(p_); return nullptr; }
        p_++;
        return n;
      }
      auto ch = element(err);
      if (!ch) return nullptr;
      n->children.push_back(std::move(ch));
    }
  }
};

}  // namespace hb
