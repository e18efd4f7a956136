// flags.hpp -- minimal --name value / --name=value parser standing in for gflags (an empty submodule in the
// reference, absent in the image).  Flag NAMES are the reference's (tool_query.cpp:26-36, tool_createdb.cpp:26-35).
#ifndef PQT_HOST_FLAGS_HPP
#define PQT_HOST_FLAGS_HPP
#include <stdlib.h>
#include <iostream>
#include <map>
#include <string>

class Flags {
 public:
  void def(const std::string& name, const std::string& dflt, const std::string& help) { vals[name] = dflt; helps[name] = help; }
  bool parse(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a == "-h" || a == "--help" || a == "-help") { usage(); return false; }
      while (!a.empty() && a[0] == '-') a.erase(0, 1);
      std::string v;
      const size_t eq = a.find('=');
      if (eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); }
      else if (i + 1 < argc) v = argv[++i];
      if (!vals.count(a)) { std::cerr << "unknown flag --" << a << std::endl; usage(); return false; }
      vals[a] = v;
    }
    return true;
  }
  std::string str(const std::string& n) const { return vals.at(n); }
  long long num(const std::string& n) const { return atoll(vals.at(n).c_str()); }
  void usage() const { for (auto& kv : helps) std::cerr << "  --" << kv.first << "  (" << vals.at(kv.first) << ")  " << kv.second << std::endl; }
 private:
  std::map<std::string, std::string> vals, helps;
};
#endif
