// Headless replay front-end: what MainController::run() does around ElasticFusion::processFrame
// (MainController.cpp:201-254) with the GUI removed.  Reads a .klg log (format of Tools/RawLogReader.cpp:29,63-109:
// int32 numFrames; per frame int64 timestamp, int32 depthSize, int32 imageSize, depth bytes (raw u16 or zlib),
// image bytes (raw RGB8, or one JPEG image decoded through the system libjpeg: include/efusion_jpeg.hpp)), replays it through
// libefusion.so and writes <log>.freiburg (+ <log>.ply with -ply).
//
//   efusion_replay -l seq.klg [-w 640 -h 480] [-cal fx fy cx cy] [-d depthCut] [-c confidence] [-t timeDelta]
//                  [-fo] [-nso] [-ftf] [-i icpWeight] [-e endFrame] [-ply] [-dev N] [-q]
//                  [-cl [-ic icpCountThresh] [-ie icpErrThresh] [-cv covThresh] [-pt photoThresh] [-ft fernThresh] [-rl]] [-icl] [-f]
//
// The flags and their defaults are MainController's (MainController.cpp:69-104: -c 10, -d 3, -i 10, -ie 4e-05, -cv 1e-05, -pt 115,
// -ft 0.3095, -t 200, -ic 40000; -rl relocalisation, -icl the ICL-NUIM conventions, -f flipped colours, -fo, -nso, -ftf, -e, -q) with one
// difference: open loop (the reference's -o) is the default here and -cl selects the closed loop (fern database, global and local
// closures, built-in optimiser).  Like the reference's run loop, the last frame of a log is not processed (RawLogReader::hasMore, see
// include/efusion_klg.hpp); -all processes every frame.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/ElasticFusion.h"
#include "../include/efusion_klg.hpp"

using efusion::KlgReader;

int main(int argc, char** argv) {
  std::string log;
  int w = 640, h = 480, timeDelta = 200, end = -1, dev = 0;
  float fx = 528, fy = 528, cx = 320, cy = 240, depthCut = 3, confidence = 10, icp = 10;
  float icpErrThresh = 4e-05f, covThresh = 1e-05f, photoThresh = 115, fernThresh = 0.3095f;   // MainController.cpp:72-75
  int icpCountThresh = 40000;                                                                 // :78
  bool fastOdom = false, so3 = true, ftf = false, ply = false, quiet = false, closeLoops = false, allFrames = false, solve = false;
  bool reloc = false, iclnuim = false, flipColors = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&](int n = 1) { if (i + n >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
    if (a == "-l") log = next();
    else if (a == "-w") w = std::atoi(next());
    else if (a == "-h") h = std::atoi(next());
    else if (a == "-cal") { fx = std::atof(next()); fy = std::atof(next()); cx = std::atof(next()); cy = std::atof(next()); }
    else if (a == "-d") depthCut = std::atof(next());
    else if (a == "-c") confidence = std::atof(next());
    else if (a == "-t") timeDelta = std::atoi(next());
    else if (a == "-i") icp = std::atof(next());
    else if (a == "-e") end = std::atoi(next());
    else if (a == "-dev") dev = std::atoi(next());
    else if (a == "-ic") icpCountThresh = std::atoi(next());
    else if (a == "-ie") icpErrThresh = (float)std::atof(next());
    else if (a == "-cv") covThresh = (float)std::atof(next());
    else if (a == "-pt") photoThresh = (float)std::atof(next());
    else if (a == "-ft") fernThresh = (float)std::atof(next());
    else if (a == "-rl") reloc = true;        // relocalisation (needs -cl to find its way back, as in the reference)
    else if (a == "-icl") iclnuim = true;     // PLY dump and raw timestamps in the destructor (ElasticFusion.cpp:108-128)
    else if (a == "-f") flipColors = true;    // LogReader::flipColors (RawLogReader.cpp:88-98)
    else if (a == "-fo") fastOdom = true;
    else if (a == "-nso") so3 = false;
    else if (a == "-ftf") ftf = true;
    else if (a == "-ply") ply = true;
    else if (a == "-q") quiet = true;
    else if (a == "-all") allFrames = true;
    else if (a == "-solve") solve = true;     // kept for old command lines: -cl always closes loops with the built-in optimiser now
    else if (a == "-o") closeLoops = false;   // the default here (the reference closes loops unless -o is given)
    else if (a == "-cl") closeLoops = true;   // local loop closure front half every frame, time window from -t
    else { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
  }
  if (log.empty()) { std::fprintf(stderr, "usage: efusion_replay -l file.klg [...]\n"); return 2; }
  try {
    Resolution::getInstance(w, h);
    Intrinsics::getInstance(fx, fy, cx, cy);
    KlgReader reader(log, w, h);
    reader.deliverLastFrame = allFrames;
    reader.flipColors = flipColors;
    // open loop: timeDelta = INT_MAX / 2 exactly as MainController does for -o (MainController.cpp:179-183)
    ElasticFusion eFusion(closeLoops ? timeDelta : 2147483647 / 2, icpCountThresh, icpErrThresh, covThresh, closeLoops, iclnuim, reloc, photoThresh,
                          confidence, depthCut, icp, fastOdom, fernThresh, so3, ftf, log, dev);
    (void)solve;
    int attempts = 0, opened = 0;
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    while (reader.hasMore() && (end < 0 || n < end)) {
      reader.getNext();
      eFusion.processFrame(reader.rgb.data(), (const uint16_t*)reader.depth.data(), reader.timestamp, 1.0f);
      if (closeLoops) { const ef_local_loop& L = eFusion.getLocalLoop(); attempts += L.attempted; opened += L.gates_ok; }
      ++n;
    }
    eFusion.synchronize();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double M[16];
    eFusion.get_T_wc().matrix(M);
    if (!quiet)
      std::printf("frames %d  %.1f fps  surfels %u  icp %g/%g  t_wc %.9g %.9g %.9g\n", n, n / dt, eFusion.getGlobalModel().lastCount(),
                  (double)eFusion.getModelToModel().lastICPError, (double)eFusion.getModelToModel().lastICPCount, M[3], M[7], M[11]);
    if (!quiet && reloc) std::printf("relocalisation: lost %d  tick %d\n", (int)eFusion.getLost(), eFusion.getTick());
    if (!quiet && closeLoops) {
      std::printf("local loop closure: attempts %d  gates open %d  deformations %d\n", attempts, opened, eFusion.getDeforms());
      const efusion::FernsView& F = eFusion.getFerns();
      std::printf("fern database: keyframes %d  global deformations %d  pose matches %d\n", (int)F.frames.size(), eFusion.getFernDeforms(),
                  (int)eFusion.getPoseMatches().size());
      // MainController.cpp:388-404,465-468: the deformation graph as the GUI draws and counts it
      const std::vector<GraphNode*>& graph = eFusion.getLocalDeformation().getGraph();
      size_t links = 0;
      double reach = 0;   // longest link
      for (size_t g = 0; g < graph.size(); g++)
        for (size_t j = 0; j < graph.at(g)->neighbours.size(); j++, links++) {
          const GraphNode* o = graph.at(graph.at(g)->neighbours.at(j));
          double d = 0;
          for (int k = 0; k < 3; k++) d += (graph.at(g)->position(k) - o->position(k)) * (graph.at(g)->position(k) - o->position(k));
          reach = d > reach ? d : reach;
        }
      std::printf("deformation graph: nodes %d  links %d  longest %.6f", (int)graph.size(), (int)links, std::sqrt(reach));
      if (!graph.empty()) std::printf("  first %.9g %.9g %.9g", graph.at(0)->position(0), graph.at(0)->position(1), graph.at(0)->position(2));
      std::printf("\n");
    }
    if (ply) eFusion.savePly();
  } catch (const std::exception& e) {
    std::fprintf(stderr, "efusion_replay: %s\n", e.what());
    return 1;
  }
  return 0;
}
