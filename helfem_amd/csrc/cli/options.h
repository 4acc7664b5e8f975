// Command-line handling of the `diatomic` and `atomic` executables: the flag names, defaults and "required" marks of the
// reference's parsers (/root/reference/src/diatomic/main.cpp:89-133, /root/reference/src/atomic/main.cpp:63-119).  Like
// the reference's cmdline parser, an option is given as `--name=value` or `--name value`; unknown options and missing
// required ones end the program with a usage message and exit status 1.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace cli {

/// arma::imat::load(raw_ascii): whitespace-separated integers, one row per line, equal row lengths
inline bool read_int_table(const char *path, std::vector<int> &tab, int &rows, int &cols) {
  FILE *f = fopen(path, "r");
  if (!f) return false;
  tab.clear();
  rows = cols = 0;
  char line[4096];
  bool ok = true;
  while (fgets(line, sizeof(line), f)) {
    int n = 0;
    char *p = line, *end = nullptr;
    while (true) {
      const long v = strtol(p, &end, 10);
      if (end == p) break;
      tab.push_back((int)v);
      n++;
      p = end;
    }
    if (!n) continue;  // blank line
    if (cols && n != cols) ok = false;
    cols = n;
    rows++;
  }
  fclose(f);
  return ok && rows > 0;
}

struct Option {
  std::string name, desc, value;
  bool required = false, given = false, is_bool = false;
};

class Parser {
 public:
  void add(const std::string &name, const std::string &desc, bool required, const std::string &def = "", bool is_bool = false) {
    Option o;
    o.name = name;
    o.desc = desc;
    o.required = required;
    o.value = def;
    o.is_bool = is_bool;
    index_[name] = opts_.size();
    opts_.push_back(o);
  }
  std::string usage(const std::string &prog) const {
    std::string u = "usage: " + prog;
    for (const Option &o : opts_)
      if (o.required) u += " --" + o.name + "=<value>";
    u += " [options] ...\noptions:\n";
    for (const Option &o : opts_) {
      char line[512];
      snprintf(line, sizeof(line), "  --%-14s %s%s%s%s\n", o.name.c_str(), o.desc.c_str(), o.required ? "" : " [=", o.required ? "" : o.value.c_str(),
               o.required ? "" : "]");
      u += line;
    }
    return u;
  }
  /// returns an error text (empty: fine)
  std::string parse(int argc, char **argv) {
    for (int i = 1; i < argc; i++) {
      std::string a = argv[i];
      if (a == "--help" || a == "-?") return "help";
      if (a.size() < 3 || a[0] != '-' || a[1] != '-') return "undefined argument: " + a;
      std::string name = a.substr(2), value;
      bool has_value = false;
      size_t eq = name.find('=');
      if (eq != std::string::npos) {
        value = name.substr(eq + 1);
        name = name.substr(0, eq);
        has_value = true;
      }
      auto it = index_.find(name);
      if (it == index_.end()) return "undefined option: --" + name;
      Option &o = opts_[it->second];
      if (!has_value) {
        if (i + 1 < argc && !(argv[i + 1][0] == '-' && argv[i + 1][1] == '-')) {
          value = argv[++i];
        } else if (o.is_bool) {
          value = "1";
        } else
          return "option needs value: --" + name;
      }
      o.value = value;
      o.given = true;
    }
    for (const Option &o : opts_)
      if (o.required && !o.given) return "need option: --" + o.name;
    return "";
  }
  /// parse_check of the reference's parser: message + usage, exit(1)
  void parse_check(int argc, char **argv) {
    std::string err = parse(argc, argv);
    if (err.empty()) return;
    if (err != "help") fprintf(stderr, "%s\n", err.c_str());
    fprintf(stderr, "%s", usage(argc ? argv[0] : "program").c_str());
    exit(err == "help" ? 0 : 1);
  }
  const std::string &str(const std::string &name) const { return opts_[index_.at(name)].value; }
  bool given(const std::string &name) const { return opts_[index_.at(name)].given; }
  /// checked conversion of one item of an option's value (the whole value, or one entry of a comma-separated list)
  static int to_int(const std::string &name, const std::string &v) {
    char *end = nullptr;
    long r = strtol(v.c_str(), &end, 10);
    if (end == v.c_str() || *end) throw std::runtime_error("option value is invalid: --" + name + "=" + v);
    return (int)r;
  }
  int integer(const std::string &name) const { return to_int(name, str(name)); }
  double real(const std::string &name) const {
    const std::string &v = str(name);
    char *end = nullptr;
    double r = strtod(v.c_str(), &end);
    if (end == v.c_str() || *end) throw std::runtime_error("option value is invalid: --" + name + "=" + v);
    return r;
  }
  bool boolean(const std::string &name) const {
    const std::string &v = str(name);
    if (v == "1" || v == "true" || v == "True") return true;
    if (v == "0" || v == "false" || v == "False" || v.empty()) return false;
    throw std::runtime_error("option value is invalid: --" + name + "=" + v);
  }

 private:
  std::vector<Option> opts_;
  std::map<std::string, size_t> index_;
};

}  // namespace cli
