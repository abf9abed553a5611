// TEST INFRASTRUCTURE — the one Pangolin utility Tools/RawLogReader.cpp uses
#pragma once
#include <cstdio>
#include <string>
namespace pangolin {
inline bool FileExists(const std::string& f) {
  if (FILE* fp = std::fopen(f.c_str(), "rb")) { std::fclose(fp); return true; }
  return false;
}
}
