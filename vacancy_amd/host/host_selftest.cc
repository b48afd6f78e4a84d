// CPU-only check of the facade's host utilities (no GPU calls): PNG decode of the bunny masks,
// TUM pose -> w2c arithmetic.  Prints values that tests/test_host.py compares with fixtures.
#include <cstdio>
#include <string>

#include "vacancy/camera.h"
#include "vacancy/image.h"

int main(int argc, char* argv[]) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  for (int i = 0; i < 6; ++i) {
    vacancy::Image1b m;
    if (!m.Load(dir + "/mask_" + vacancy::zfill(i) + ".png")) return 1;
    unsigned long long sum = 0;
    for (unsigned char p : m.data()) sum += p;
    std::printf("MASK %d %d %d %llu\n", i, m.width(), m.height(), sum);
  }
  // view 2 of tumpose.txt
  Eigen::Translation3d t;
  t.x() = 710.836121; t.y() = -48.510956; t.z() = 31.836634;
  Eigen::Quaterniond q;
  q.x() = 0.0; q.y() = -0.707107; q.z() = 0.0; q.w() = 0.707107;
  vacancy::PinholeCamera cam(320, 240, t * q, Eigen::Vector2f(159.3f, 127.65f), Eigen::Vector2f(258.65f, 258.25f));
  const Eigen::Affine3f w2c = cam.w2c().cast<float>();
  std::printf("W2C");
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) std::printf(" %.9g", w2c.linear()(i, j));
    std::printf(" %.9g", w2c.translation()[i]);
  }
  std::printf("\n");
  vacancy::PinholeCamera fov(1280, 720, 60.0f);
  std::printf("FOCAL %.9g %.9g %.9g\n", fov.focal_length()[0], fov.principal_point()[0], fov.principal_point()[1]);
  return 0;
}
