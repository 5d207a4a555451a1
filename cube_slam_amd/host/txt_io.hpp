// txt_io.hpp -- header-only C++ twins of the reference's text readers, without Eigen.
//
// read_all_number_txt / read_obj_detection_txt / read_obj_detection2_txt follow
// detect_3d_cuboid/src/matrix_utils.cpp:195-314: blank lines skipped, a row is parsed number by number until the first token
// that is not a number, the column count is the caller's (10 by default, :209-210), missing trailing columns are 0.
// The adapters in INTEGRATION.md map NumMat onto Eigen::MatrixXd (row-major data, rows x cols).
#pragma once
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace cubeslam {

struct NumMat {
    int rows = 0, cols = 0;
    std::vector<double> data; // row-major
    double &operator()(int r, int c) { return data[(size_t)r * cols + c]; }
    double operator()(int r, int c) const { return data[(size_t)r * cols + c]; }
};

// returns false when the file cannot be opened (the reference prints "ERROR!!! Cannot read txt file" and returns false)
inline bool read_all_number_txt(const std::string &txt_file_name, NumMat &m, int cols = 10) {
    std::ifstream f(txt_file_name.c_str());
    if (!f) return false;
    m.rows = 0; m.cols = cols; m.data.clear();
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        std::stringstream ss(line);
        m.data.resize((size_t)(m.rows + 1) * cols, 0.0);
        double t; int c = 0;
        while (c < cols && ss >> t) m.data[(size_t)m.rows * cols + c++] = t;
        m.rows++;
    }
    return true;
}
// class name first, numbers after it (:235-270)
inline bool read_obj_detection_txt(const std::string &txt_file_name, NumMat &m, std::vector<std::string> &all_strings, int cols = 10) {
    std::ifstream f(txt_file_name.c_str());
    if (!f) return false;
    all_strings.clear();
    m.rows = 0; m.cols = cols; m.data.clear();
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        std::stringstream ss(line);
        std::string classname;
        ss >> classname;
        all_strings.push_back(classname);
        m.data.resize((size_t)(m.rows + 1) * cols, 0.0);
        double t; int c = 0;
        while (c < cols && ss >> t) m.data[(size_t)m.rows * cols + c++] = t;
        m.rows++;
    }
    return true;
}
// `cols` numbers first, class name after them (:272-313)
inline bool read_obj_detection2_txt(const std::string &txt_file_name, NumMat &m, std::vector<std::string> &all_strings, int cols) {
    std::ifstream f(txt_file_name.c_str());
    if (!f) return false;
    all_strings.clear();
    m.rows = 0; m.cols = cols; m.data.clear();
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        std::stringstream ss(line);
        m.data.resize((size_t)(m.rows + 1) * cols, 0.0);
        double t; int c = 0;
        while (c < cols && ss >> t) m.data[(size_t)m.rows * cols + c++] = t;
        ss.clear();
        std::string classname;
        ss >> classname;
        all_strings.push_back(classname);
        m.rows++;
    }
    return true;
}
// LSD dump written by line_lbd/src/detect_lines.cpp:85-96 (x1 y1 x2 y2, tab separated)
inline bool write_edge_txt(const std::string &txt_file_name, const float *lines, int n) {
    std::ofstream f(txt_file_name.c_str());
    if (!f) return false;
    for (int j = 0; j < n; j++) f << lines[j * 4] << "\t" << lines[j * 4 + 1] << "\t" << lines[j * 4 + 2] << "\t" << lines[j * 4 + 3] << std::endl;
    return true;
}

} // namespace cubeslam
