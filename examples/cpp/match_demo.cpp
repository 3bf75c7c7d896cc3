// match_demo.cpp -- C++ consumer of include/rgbdfe.hpp, written the way the reference's call site
// (GraphManager::nodeComparisons, graph_manager.cpp:531-583) would use it.
//   match_demo <nodes.bin> [multi]      "multi": two device contexts behind one handle (rgbdfe_create_multi, device 0 twice)
// nodes.bin: int32 n_nodes, then per node: int32 n, n*32 bytes descriptors, n*4 floats xyz1.
// Prints one JSON line per (last node, earlier node) pair.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "rgbdfe.hpp"

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: match_demo nodes.bin\n"); return 2; }
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 2; }
  int32_t n_nodes = 0;
  if (std::fread(&n_nodes, 4, 1, f) != 1) return 2;
  rgbdfe_config cfg = rgbdslam::FrontEnd::defaultConfig();
  cfg.max_nodes = n_nodes + 12;  // + the frame nodes and the SIFT nodes of the optional sections
  cfg.max_keypoints = 2048;
  cfg.max_pairs_per_batch = 64;
  const bool multi = argc > 2 && std::string(argv[2]) == "multi";
  rgbdslam::FrontEnd fe = multi ? rgbdslam::FrontEnd(cfg, std::vector<int32_t>{0, 0}) : rgbdslam::FrontEnd(cfg);
  std::vector<std::unique_ptr<rgbdslam::Node>> graph;
  for (int i = 0; i < n_nodes; ++i) {
    int32_t n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return 2;
    std::vector<uint8_t> desc((size_t)n * 32);
    std::vector<float> xyz((size_t)n * 4);
    if (n && (std::fread(desc.data(), 32, (size_t)n, f) != (size_t)n || std::fread(xyz.data(), 16, (size_t)n, f) != (size_t)n)) return 2;
    graph.emplace_back(new rgbdslam::Node(fe, i, desc.data(), xyz.data(), n));
  }
  std::fclose(f);
  rgbdslam::GraphManager gm(fe);
  const rgbdslam::Node* new_node = graph.back().get();
  std::vector<const rgbdslam::Node*> nodes_to_comp;
  for (int i = 0; i + 1 < n_nodes; ++i) nodes_to_comp.push_back(graph[i].get());
  const std::vector<rgbdslam::MatchingResult> results = gm.nodeComparisons(new_node, nodes_to_comp);
  for (const rgbdslam::MatchingResult& mr : results) {
    std::printf("{\"id1\": %d, \"id2\": %d, \"n_all\": %zu, \"n_inl\": %zu, \"rmse\": %.9g, \"info\": %.17g, \"T\": [",
                mr.edge.id1, mr.edge.id2, mr.all_matches.size(), mr.inlier_matches.size(), mr.rmse, mr.edge.informationScale);
    for (int i = 0; i < 16; ++i) std::printf("%s%.9g", i ? ", " : "", mr.final_trafo[i]);
    std::printf("]}\n");
  }
  // the serial call site (graph_manager.cpp:466): one pair through Node::matchNodePair
  const rgbdslam::MatchingResult one = new_node->matchNodePair(graph.front().get());
  // candidate selection for the next node (graph_manager.cpp:204-324): the earlier nodes as a chain of keyframes
  for (int i = 0; i + 1 < n_nodes; ++i) {
    gm.nodeAdded(graph[i].get(), /*vertex_id=*/i, /*keyframe=*/true);
    if (i) gm.edgeAdded(i, i - 1);
  }
  const std::vector<int> targets = gm.getPotentialEdgeTargetsWithDijkstra(new_node, 2, 1, 1, -1, false, 3, nullptr, nullptr, 7u);
  std::printf("{\"candidates\": [");
  for (size_t i = 0; i < targets.size(); ++i) std::printf("%s%d", i ? ", " : "", targets[i]);
  std::printf("]}\n");
  std::printf("{\"single_id1\": %d, \"single_n_inl\": %zu}\n", one.edge.id1, one.inlier_matches.size());
  // place recognition (loop_closing.cpp:190-277): the earlier nodes ranked by descriptor votes
  const std::vector<int> ranked = gm.getNeighbours(new_node, nodes_to_comp, /*neighbour_cnt=*/2, /*max_out=*/n_nodes);
  std::printf("{\"devices\": %d, \"neighbours\": [", fe.deviceCount());
  for (size_t i = 0; i < ranked.size(); ++i) std::printf("%s%d", i ? ", " : "", ranked[i]);
  std::printf("]}\n");
  // SIFT nodes (matcher_type == "SIFTGPU"): a file of 128-d float descriptors, the last node against the others
  if (argc > 4) {
    std::FILE* s = std::fopen(argv[4], "rb");
    if (!s) { std::perror("open sift"); return 2; }
    int32_t ns = 0;
    if (std::fread(&ns, 4, 1, s) != 1) return 2;
    std::vector<std::unique_ptr<rgbdslam::Node>> sg;
    for (int i = 0; i < ns; ++i) {
      int32_t n = 0;
      if (std::fread(&n, 4, 1, s) != 1) return 2;
      std::vector<float> desc((size_t)n * 128), xyz((size_t)n * 4);
      if (std::fread(desc.data(), 512, (size_t)n, s) != (size_t)n || std::fread(xyz.data(), 16, (size_t)n, s) != (size_t)n) return 2;
      sg.emplace_back(new rgbdslam::Node(fe, 100 + i, desc.data(), xyz.data(), n, rgbdslam::Node::SiftDescriptors()));
    }
    std::fclose(s);
    std::vector<const rgbdslam::Node*> older;
    for (int i = 0; i + 1 < ns; ++i) older.push_back(sg[(size_t)i].get());
    const std::vector<rgbdslam::MatchingResult> rs = gm.nodeComparisons(sg.back().get(), older);
    const rgbdslam::MatchingResult one_s = sg.back()->matchNodePair(sg.front().get());
    for (const rgbdslam::MatchingResult& mr : rs) {
      double dsum = 0;
      for (const rgbdslam::DMatch& m : mr.all_matches) dsum += m.distance;
      std::printf("{\"sift_id1\": %d, \"sift_id2\": %d, \"sift_n_all\": %zu, \"sift_n_inl\": %zu, \"sift_dist_sum\": %.9g}\n", mr.edge.id1,
                  mr.edge.id2, mr.all_matches.size(), mr.inlier_matches.size(), dsum);
    }
    std::printf("{\"sift_single_n_inl\": %zu}\n", one_s.inlier_matches.size());
  }
  // the depth-image Node constructor (node.cpp:139-210): two frames of a file written by the test, matched to each other
  if (argc > 3 && std::string(argv[3]) != "-") {
    std::FILE* g = std::fopen(argv[3], "rb");
    if (!g) { std::perror("open frames"); return 2; }
    int32_t hdr[2];
    double K[4];
    if (std::fread(hdr, 4, 2, g) != 2 || std::fread(K, 8, 4, g) != 4) return 2;
    const int rows = hdr[0], cols = hdr[1];
    rgbdfe_detector_configure(fe.get(), 1000, 3, 5);
    std::vector<std::unique_ptr<rgbdslam::Node>> fr;
    for (int i = 0; i < 2; ++i) {
      std::vector<uint8_t> gray((size_t)rows * cols), mask((size_t)rows * cols);
      std::vector<float> depth((size_t)rows * cols);
      if (std::fread(gray.data(), 1, gray.size(), g) != gray.size() || std::fread(mask.data(), 1, mask.size(), g) != mask.size() ||
          std::fread(depth.data(), 4, depth.size(), g) != depth.size()) return 2;
      fr.emplace_back(new rgbdslam::Node(fe, n_nodes + i, gray.data(), mask.data(), depth.data(), rows, cols, K[0], K[1], K[2],
                                         K[3], 1.0, 1000));
    }
    // optionally one organised cloud for the second frame: the point-cloud constructor (node.cpp:218-369)
    std::vector<float> cloud((size_t)rows * cols * 4);
    const bool have_cloud = std::fread(cloud.data(), 4, cloud.size(), g) == cloud.size();
    std::fclose(g);
    if (have_cloud) {
      std::FILE* g2 = std::fopen(argv[3], "rb");
      std::fseek(g2, 8 + 32 + (long)((size_t)rows * cols * 6), SEEK_SET);  // header, K, frame 0
      std::vector<uint8_t> gray((size_t)rows * cols), mask((size_t)rows * cols);
      if (std::fread(gray.data(), 1, gray.size(), g2) != gray.size() || std::fread(mask.data(), 1, mask.size(), g2) != mask.size()) return 2;
      std::fclose(g2);
      rgbdslam::Node pc(fe, n_nodes + 2, gray.data(), mask.data(), cloud.data(), rows, cols, 3.5, 1000,
                        rgbdslam::Node::FromPointCloud());
      unsigned long long s2 = 0;
      for (uint8_t b : pc.feature_descriptors_) s2 = s2 * 131 + b;
      float zsum = 0;
      for (int i = 0; i < pc.featureCount(); ++i) zsum += pc.feature_locations_3d_[(size_t)i * 4 + 2];
      std::printf("{\"cloud_features\": %d, \"cloud_desc_hash\": %llu, \"cloud_zsum\": %.9g}\n", pc.featureCount(), s2, zsum);
    }
    const rgbdslam::MatchingResult mr = fr[1]->matchNodePair(fr[0].get());
    unsigned long long sum = 0;
    for (uint8_t b : fr[1]->feature_descriptors_) sum = sum * 131 + b;
    std::printf("{\"frame_features\": [%d, %d], \"frame_edge\": [%d, %d], \"frame_inliers\": %zu, \"desc_hash\": %llu}\n",
                fr[0]->featureCount(), fr[1]->featureCount(), mr.edge.id1, mr.edge.id2, mr.inlier_matches.size(), sum);
    // the same two frames as SiftGPU nodes (feature_detector_type == "SIFTGPU": SiftGPUWrapper::detect -> projectTo3DSiftGPU),
    // matched with the SIFTGPU matcher branch
    {
      std::FILE* g3 = std::fopen(argv[3], "rb");
      std::fseek(g3, 8 + 32, SEEK_SET);
      std::vector<std::unique_ptr<rgbdslam::Node>> sn;
      for (int i = 0; i < 2; ++i) {
        std::vector<uint8_t> gray((size_t)rows * cols), mask((size_t)rows * cols);
        std::vector<float> depth((size_t)rows * cols);
        if (std::fread(gray.data(), 1, gray.size(), g3) != gray.size() || std::fread(mask.data(), 1, mask.size(), g3) != mask.size() ||
            std::fread(depth.data(), 4, depth.size(), g3) != depth.size()) return 2;
        sn.emplace_back(new rgbdslam::Node(fe, 200 + i, gray.data(), depth.data(), rows, cols, K[0], K[1], K[2], K[3], 1.0, 1000,
                                           rgbdslam::Node::SiftGPU()));
      }
      std::fclose(g3);
      const rgbdslam::MatchingResult ms = sn[1]->matchNodePair(sn[0].get());
      double dsum = 0;
      for (float v : sn[1]->siftgpu_descriptors_) dsum += v;
      std::printf("{\"siftgpu_features\": [%d, %d], \"siftgpu_edge\": [%d, %d], \"siftgpu_inliers\": %zu, \"siftgpu_desc_sum\": %.9g}\n",
                  sn[0]->featureCount(), sn[1]->featureCount(), ms.edge.id1, ms.edge.id2, ms.inlier_matches.size(), dsum);
    }
  }
  return 0;
}
