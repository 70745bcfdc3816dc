/*
 * ParameterFile.hpp - the ".param" driver input: an indentation-based YAML
 * subset flattened to "Block:key" -> string, typed getters with defaults,
 * unit-aware getters, and a record of every value that was actually used.
 *
 * Host-side mirror of the reference's ParameterFile / YAMLDictionary
 * (src/ParameterFile.hpp:103-170, src/YAMLDictionary.hpp:177-268 parser,
 * :389-520 getters, :270-380 print_contents): same grammar, same defaulting
 * rules ("default value" marker), same "<file>.used-values" dump format.
 */
#ifndef CMI_HOST_PARAMETERFILE_HPP
#define CMI_HOST_PARAMETERFILE_HPP

#include "Units.hpp"

#include <algorithm>
#include <array>
#include <cstdint>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace cmi {

class ParameterFile {
  std::map<std::string, std::string> _dictionary;
  std::map<std::string, std::string> _used;
  std::string _filename;

  static std::string strip(const std::string &s) {
    const size_t a = s.find_first_not_of(" \t");
    if (a == std::string::npos)
      return "";
    const size_t b = s.find_last_not_of(" \t");
    return s.substr(a, b - a + 1);
  }

  static std::string number_to_string(double v) {
    /* Utilities::to_string<double>: default ostream formatting */
    std::stringstream ss;
    ss << v;
    return ss.str();
  }

  void parse(std::istream &stream) {
    std::string line;
    std::vector<std::string> groups;
    std::vector<size_t> levels;
    while (std::getline(stream, line)) {
      const size_t first = line.find_first_not_of(" \t");
      if (first == std::string::npos || line[first] == '#')
        continue; /* empty or comment line */
      const size_t hash = line.find('#');
      if (hash != std::string::npos)
        line = line.substr(0, hash);
      const size_t colon = line.find(':');
      if (colon == std::string::npos)
        throw ParameterError("Error while parsing line \"" + line +
                             "\": no ':' found!");
      const std::string name = strip(line.substr(0, colon));
      const std::string value = strip(line.substr(colon + 1));
      const size_t indent = first;
      std::string key;
      if (indent > 0) {
        if (!levels.empty()) {
          if (indent > levels.back()) {
            levels.push_back(indent);
          } else {
            while (!levels.empty() && indent < levels.back()) {
              levels.pop_back();
              groups.pop_back();
            }
          }
        } else {
          levels.push_back(indent);
        }
        if (levels.size() != groups.size())
          throw ParameterError(
              "Line has a different indentation than expected: \"" + line +
              "\"!");
        if (value.empty()) {
          groups.push_back(name);
        } else {
          for (const auto &g : groups)
            key += g + ":";
          key += name;
        }
      } else {
        if (groups.size() != levels.size())
          throw ParameterError("Wrong formatting!");
        levels.clear();
        groups.clear();
        key = name;
        if (value.empty())
          groups.push_back(key);
      }
      if (!value.empty())
        _dictionary[key] = value;
    }
  }

  /* YAMLDictionary::get_value<std::string>(key, default), :650-670 */
  std::string raw(const std::string &key, const std::string &fallback) {
    auto it = _dictionary.find(key);
    std::string s;
    if (it == _dictionary.end() || it->second == "default value") {
      _dictionary[key] = "default value";
      s = fallback;
    } else {
      s = it->second;
    }
    _used[key] = s;
    return s;
  }

  /* Utilities::string_to_integer, src/Utilities.hpp:118-214: digits with an
   * optional e<digits> exponent ("1e8"); parsing stops at the first other
   * character */
  static long long to_integer(const std::string &value) {
    size_t idx = 0;
    while (idx < value.size() && value[idx] == ' ')
      ++idx;
    if (idx == value.size())
      throw ParameterError("String does not contain an integer: \"" + value +
                           "\"!");
    long long sign = 1;
    if (value[idx] == '-') {
      sign = -1;
      ++idx;
    }
    long long ivalue = 0;
    if (value[idx] == '0' && idx + 1 < value.size() &&
        (value[idx + 1] == 'x' || value[idx + 1] == 'X')) {
      idx += 2;
      while (idx < value.size() && isxdigit((unsigned char)value[idx])) {
        const char c = (char)tolower(value[idx]);
        ivalue = ivalue * 16 + (isdigit((unsigned char)c) ? c - '0'
                                                          : 10 + c - 'a');
        ++idx;
      }
    } else {
      while (idx < value.size() && isdigit((unsigned char)value[idx])) {
        ivalue = ivalue * 10 + (value[idx] - '0');
        ++idx;
      }
      if (idx < value.size() && (value[idx] == 'e' || value[idx] == 'E')) {
        ++idx;
        long long exponent = 0;
        while (idx < value.size() && isdigit((unsigned char)value[idx])) {
          exponent = exponent * 10 + (value[idx] - '0');
          ++idx;
        }
        for (long long i = 0; i < exponent; ++i)
          ivalue *= 10;
      }
    }
    return ivalue * sign;
  }

  /* Utilities::convert<bool>, src/Utilities.hpp:487-515 */
  static bool to_bool(const std::string &value) {
    std::string v = strip(value);
    std::transform(v.begin(), v.end(), v.begin(), ::tolower);
    if (v == "true" || v == "yes" || v == "on" || v == "y")
      return true;
    if (v == "false" || v == "no" || v == "off" || v == "n")
      return false;
    throw ParameterError("Error converting \"" + v + "\" to a boolean value!");
  }

  /* Utilities::split_string, src/Utilities.hpp:96-107: "[a, b, c]" */
  static std::array<std::string, 3> split3(const std::string &value) {
    std::array<std::string, 3> out;
    size_t p1 = value.find('[') + 1;
    size_t p2 = value.find(',', p1);
    out[0] = value.substr(p1, p2 - p1);
    p1 = p2 + 1;
    p2 = value.find(',', p1);
    out[1] = value.substr(p1, p2 - p1);
    p1 = p2 + 1;
    p2 = value.find(']', p1);
    out[2] = value.substr(p1, p2 - p1);
    return out;
  }

public:
  ParameterFile() {}
  explicit ParameterFile(const std::string &filename) : _filename(filename) {
    std::ifstream file(filename);
    if (!file)
      throw ParameterError("Failed to open parameter file \"" + filename +
                           "\"");
    parse(file);
  }
  explicit ParameterFile(std::istream &stream) { parse(stream); }

  const std::string &filename() const { return _filename; }
  bool has_value(const std::string &key) const {
    return _dictionary.count(key) > 0;
  }
  void add_value(const std::string &key, const std::string &value) {
    _dictionary[key] = value;
    _used[key] = value;
  }

  std::string get_string(const std::string &key, const std::string &fallback) {
    return raw(key, fallback);
  }
  /* path relative to the parameter file's folder (ParameterFile::get_filename) */
  std::string get_filename(const std::string &key) {
    auto it = _dictionary.find(key);
    if (it == _dictionary.end())
      throw ParameterError("Parameter \"" + key + "\" not found!");
    _used[key] = it->second;
    std::string name = it->second;
    if (!name.empty() && name[0] != '/') {
      const size_t slash = _filename.find_last_of('/');
      if (slash != std::string::npos)
        name = _filename.substr(0, slash + 1) + name;
    }
    return name;
  }
  double get_double(const std::string &key, double fallback) {
    const std::string s = raw(key, "");
    const double v = s.empty() ? fallback : std::stod(s);
    _used[key] = number_to_string(v);
    return v;
  }
  long long get_integer(const std::string &key, long long fallback) {
    const std::string s = raw(key, "");
    const long long v = s.empty() ? fallback : to_integer(s);
    _used[key] = std::to_string(v);
    return v;
  }
  bool get_bool(const std::string &key, bool fallback) {
    const std::string s = raw(key, "");
    const bool v = s.empty() ? fallback : to_bool(s);
    _used[key] = v ? "true" : "false";
    return v;
  }
  std::array<long long, 3> get_integer_vector(
      const std::string &key, const std::array<long long, 3> &fallback) {
    const std::string s = raw(key, "");
    std::array<long long, 3> v = fallback;
    if (!s.empty()) {
      const auto parts = split3(s);
      for (int i = 0; i < 3; ++i)
        v[i] = to_integer(parts[i]);
    }
    _used[key] = "[" + std::to_string(v[0]) + ", " + std::to_string(v[1]) +
                 ", " + std::to_string(v[2]) + "]";
    return v;
  }
  std::array<bool, 3> get_bool_vector(const std::string &key,
                                      const std::array<bool, 3> &fallback) {
    const std::string s = raw(key, "");
    std::array<bool, 3> v = fallback;
    if (!s.empty()) {
      const auto parts = split3(s);
      for (int i = 0; i < 3; ++i)
        v[i] = to_bool(parts[i]);
    }
    auto b = [](bool x) { return std::string(x ? "true" : "false"); };
    _used[key] = "[" + b(v[0]) + ", " + b(v[1]) + ", " + b(v[2]) + "]";
    return v;
  }
  /* get_physical_value<QUANTITY>(key, "10. pc") */
  double get_physical_value(Quantity q, const std::string &key,
                            const std::string &fallback) {
    const std::string s = raw(key, fallback);
    const auto vu = split_value(s);
    const double v = to_SI(q, vu.first, vu.second);
    _used[key] = number_to_string(v) + " " + SI_unit_name(q);
    return v;
  }
  std::array<double, 3> get_physical_vector(Quantity q, const std::string &key,
                                            const std::string &fallback) {
    const std::string s = raw(key, fallback);
    const auto parts = split3(s);
    std::array<double, 3> v;
    std::string used = "[";
    for (int i = 0; i < 3; ++i) {
      const auto vu = split_value(parts[i]);
      v[i] = to_SI(q, vu.first, vu.second);
      used += number_to_string(v[i]) + " " + SI_unit_name(q);
      if (i < 2)
        used += ", ";
    }
    _used[key] = used + "]";
    return v;
  }

  /* YAMLDictionary::print_contents(stream, used_values = true), :270-380:
   * "key: used value # (value in the file)" grouped and indented by block */
  void print_contents(std::ostream &stream) const {
    std::vector<std::string> open;
    for (const auto &kv : _dictionary) {
      std::vector<std::string> groups;
      size_t s = 0, c = kv.first.find(':');
      while (c != std::string::npos) {
        groups.push_back(kv.first.substr(s, c - s));
        s = c + 1;
        c = kv.first.find(':', s);
      }
      size_t common = 0;
      while (common < open.size() && common < groups.size() &&
             open[common] == groups[common])
        ++common;
      open.resize(common);
      std::string indent(2 * common, ' ');
      for (size_t j = common; j < groups.size(); ++j) {
        open.push_back(groups[j]);
        stream << indent << groups[j] << ":\n";
        indent += "  ";
      }
      const auto u = _used.find(kv.first);
      stream << indent << kv.first.substr(s) << ": "
             << (u != _used.end() ? u->second : std::string("value not used"))
             << " # (" << kv.second << ")\n";
    }
  }

  const std::map<std::string, std::string> &used_values() const {
    return _used;
  }
};

} // namespace cmi

#endif
