#pragma once
// declaration-only stand-in (see README.md): the grid_map names the class header and the binding mention
#include <string>
#include <utility>
#include <vector>
namespace grid_map {
struct Position { double x() const; double y() const; };
struct Index { Index(int row, int col); int operator()(int k) const; };
class Matrix { // Eigen::MatrixXf
  public:
    float *data();
    const float *data() const;
    void resize(long rows, long cols);
    long rows() const;
    long cols() const;
};
class GridMap {
  public:
    const Position &getPosition() const;
    bool exists(const std::string &layer) const;
    void add(const std::string &layer, const double value);
    Matrix &operator[](const std::string &layer);
};
}
