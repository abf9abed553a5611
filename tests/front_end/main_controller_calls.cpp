// TEST INFRASTRUCTURE (tests/test_front_end_compiles.py) — every use the reference's front-end makes of class ElasticFusion
// (MainController.cpp:178-194 construction, :203-254 the run loop, :262-500 statistics / GUI feedback / setters, :520 savePly), written
// out as the reference writes it and compiled against include/ElasticFusion.h with -fsyntax-only.  The test that drives this file first
// takes the census of `eFusion->member` uses from /root/reference/MainController.cpp itself and fails when the reference uses a member
// that is neither exercised here nor on the allow-list of GL-typed members.
#include <cmath>
#include <cstdint>
#include <sstream>
#include <string>
#include <vector>

#include <ElasticFusion.h>

struct LogReaderLike {   // the members of Tools/LogReader.h the run loop hands to processFrame
  unsigned char* rgb;
  unsigned short* depth;
  int64_t timestamp;
};

int front_end(int argc, char**) {
  // MainController.cpp:37-41
  Resolution::getInstance(640, 480);
  Intrinsics::getInstance(528, 528, 320, 240);
  // MainController.cpp:178-194: the sixteen constructor arguments, in the reference's order
  const int timeDelta = 200, icpCountThresh = 35000;
  const float icpErrThresh = 5e-05f, covThresh = 1e-05f, photoThresh = 115, confidence = 10, depth = 3, icp = 10, fernThresh = 0.3095f;
  const bool openLoop = argc > 1, iclnuim = false, reloc = false, fastOdom = false, so3 = true, frameToFrameRGB = false;
  std::string logFile = "log.klg";
  ElasticFusion* eFusion = new ElasticFusion(openLoop ? std::numeric_limits<int>::max() / 2 : timeDelta, icpCountThresh, icpErrThresh, covThresh,
                                             !openLoop, iclnuim, reloc, photoThresh, confidence, depth, icp, fastOdom, fernThresh, so3,
                                             frameToFrameRGB, logFile);
  LogReaderLike reader{nullptr, nullptr, 0};
  LogReaderLike* logReader = &reader;
  const int start = 1, end = 100, framesToSkip = 0;
  float weightMultiplier = framesToSkip + 1;
  // :203-254
  while (!(eFusion->getTick() == end)) {
    if (eFusion->getTick() < end) {
      if (eFusion->getTick() < start) {
        eFusion->setTick(start);
      }
      eFusion->setTick(eFusion->getTick() + framesToSkip);
      Sophus::SE3d* currentPose = 0;
      eFusion->processFrame(logReader->rgb, logReader->depth, logReader->timestamp, weightMultiplier, currentPose);
      Sophus::SE3d T_wc_gt;
      eFusion->processFrame(logReader->rgb, logReader->depth, logReader->timestamp, weightMultiplier, &T_wc_gt);
    } else {
      eFusion->predict();
    }
    // :262, :313 (the reference continues with .matrix().cast<float>() on Sophus' type)
    const Sophus::SE3d& T_wc = eFusion->get_T_wc();
#ifdef EFUSION_USE_SOPHUS
    const auto T_wc_matrix = eFusion->get_T_wc().matrix();   // Eigen::Matrix4d, exactly the reference's expression
    (void)T_wc_matrix;
#endif
    (void)T_wc;
    // :294-310
    std::stringstream stri, stre;
    stri << eFusion->getModelToModel().lastICPCount;
    stre << (std::isnan(eFusion->getModelToModel().lastICPError) ? 0 : eFusion->getModelToModel().lastICPError);
    // :348-363 (arguments of the draw calls)
    const float c = eFusion->getConfidenceThreshold();
    const int tk = eFusion->getTick(), td = eFusion->getTimeDelta();
    (void)c; (void)tk; (void)td;
    // :369-416
    if (eFusion->getLost()) {
    }
    for (size_t i = 0; i < eFusion->getFerns().frames.size(); i++) {
      if ((int)i == eFusion->getFerns().lastClosest) continue;
      const auto M_fern = eFusion->getFerns().frames.at(i)->T_wc.cast<float>().matrix();   // :383, as written (the view's entries answer to -> like Ferns::Frame*)
      (void)M_fern(0, 3);
    }
    const std::vector<GraphNode*>& graph = eFusion->getLocalDeformation().getGraph();        // :389-403, as written
    for (size_t i = 0; i < graph.size(); i++) {
      double x = graph.at(i)->position(0) + graph.at(i)->position(1) + graph.at(i)->position(2);
      for (size_t j = 0; j < graph.at(i)->neighbours.size(); j++) x += graph.at(graph.at(i)->neighbours.at(j))->position(0);
      (void)x;
    }
    const std::vector<PoseMatch>& poseMatches = eFusion->getPoseMatches();                    // :416-441, as written
    int maxDiff = 0;
    for (size_t i = 0; i < poseMatches.size(); i++) {
      if (poseMatches.at(i).secondId - poseMatches.at(i).firstId > maxDiff) maxDiff = poseMatches.at(i).secondId - poseMatches.at(i).firstId;
      if (poseMatches.at(i).fern) {
      }
      for (size_t j = 0; j < poseMatches.at(i).constraints.size(); j++) {
        const double d = poseMatches.at(i).constraints.at(j).sourcePoint(0) + poseMatches.at(i).constraints.at(j).sourcePoint(1) +
                         poseMatches.at(i).constraints.at(j).sourcePoint(2) - poseMatches.at(i).constraints.at(j).targetPoint(0) -
                         poseMatches.at(i).constraints.at(j).targetPoint(1) - poseMatches.at(i).constraints.at(j).targetPoint(2);
        (void)d;
      }
    }
    // :455-458 draw from getIndexMap()'s GL textures; what a GL-free front-end reads instead are the same images as host copies
    const std::vector<uint8_t> modelImg = eFusion->getIndexMap().image();
    (void)modelImg;
    // :461-486
    std::stringstream strs, strs2, strs3, strs4, strs5, strs6;
    strs << eFusion->getGlobalModel().lastCount();
    strs2 << eFusion->getLocalDeformation().getGraph().size();
    strs3 << eFusion->getFerns().frames.size();
    strs4 << eFusion->getDeforms();
    strs5 << eFusion->getTick() << "/" << 100;
    strs6 << eFusion->getFernDeforms();
    // :493-500
    const bool flag = true;
    const float value = 1.0f;
    eFusion->setRgbOnly(flag);
    eFusion->setPyramid(flag);
    eFusion->setFastOdom(flag);
    eFusion->setConfidenceThreshold(value);
    eFusion->setDepthCutoff(value);
    eFusion->setIcpWeight(value);
    eFusion->setSo3(flag);
    eFusion->setFrameToFrameRGB(flag);
  }
  // :520
  eFusion->savePly();
  delete eFusion;
  return 0;
}
