// STUB (see README.txt) of include/slam6d/point.h: declarations only.
#ifndef STUB_POINT_H
#define STUB_POINT_H
class Point {
public:
  Point();
  Point(double _x, double _y, double _z);
  double x, y, z;
};
#endif
