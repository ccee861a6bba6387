#pragma once
// declaration-only stand-in (see README.md): the grid_map names the class header and the binding mention
#include <string>
#include <utility>
#include <vector>
namespace grid_map {
struct Position { Position(); Position(double x, double y); double x() const; double y() const; };
struct Length { Length(double x, double y); double x() const; double y() const; };
struct Size { int operator()(int k) const; };
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
    explicit GridMap(const std::vector<std::string> &layers);
    void setFrameId(const std::string &frameId);
    void setGeometry(const Length &length, const double resolution, const Position &position);
    void setPosition(const Position &position);
    const Length &getLength() const;
    const Size &getSize() const;
    const Position &getPosition() const;
    bool exists(const std::string &layer) const;
    void add(const std::string &layer, const double value);
    Matrix &operator[](const std::string &layer);
};
}
