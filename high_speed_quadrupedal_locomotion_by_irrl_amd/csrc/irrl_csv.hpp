// irrl_csv.hpp -- host reader of the reference-trajectory CSV (VectorizedEnvironment.hpp:33-76 `readCSV_m`: one row per
// line, comma separated, no header, atof on every field).  Rows shorter than the first one are zero-padded.
#pragma once
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

namespace irrl_host {

inline bool read_csv_f32(const std::string &path, std::vector<float> &data, int &rows, int &cols, std::string &err) {
  std::ifstream in(path);
  rows = cols = 0;
  data.clear();
  if (!in.is_open()) { err = "Can Not Load Parameter File of " + path; return false; }   // the reference's message (VEC:72)
  std::string line;
  std::vector<std::vector<float>> all;
  while (std::getline(in, line, '\n')) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) continue;
    std::vector<float> row;
    size_t start = 0;
    for (size_t i = 0; i <= line.size(); i++) {
      if (i == line.size() || line[i] == ',') {
        row.push_back((float)std::atof(line.substr(start, i - start).c_str()));
        start = i + 1;
      }
    }
    if (all.empty()) cols = (int)row.size();
    row.resize((size_t)cols, 0.0f);
    all.push_back(row);
  }
  rows = (int)all.size();
  if (rows == 0) { err = "empty reference trajectory file " + path; return false; }
  data.reserve((size_t)rows * cols);
  for (auto &r : all) data.insert(data.end(), r.begin(), r.end());
  return true;
}

}  // namespace irrl_host
