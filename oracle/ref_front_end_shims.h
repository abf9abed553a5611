// TEST INFRASTRUCTURE — force-included into the reference's front-end sources by `make reffrontend` (see ref_front_end_bridge.cpp)
#pragma once
// Core/Shaders/Resize.h: a GL resize pass MainController constructs (MainController.cpp:118-122), deletes (:142-144) and never uses
class Resize {
 public:
  Resize(int, int, int, int) {}
};
