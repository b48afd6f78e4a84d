// Triangle mesh container filled by VoxelCarver::ExtractIsoSurface.  Subset of the reference's
// include/vacancy/mesh.h that the carving path uses: Clear, set_vertices, set_vertex_indices,
// accessors and the ASCII PLY writer (byte-compatible with reference mesh.cc:583-631).
#pragma once

#include <string>
#include <vector>

#include "vacancy/common.h"

namespace vacancy {

class Mesh {
 public:
  void Clear() { vertices_.clear(); vertex_colors_.clear(); vertex_indices_.clear(); }
  const std::vector<Eigen::Vector3f>& vertices() const { return vertices_; }
  const std::vector<Eigen::Vector3f>& vertex_colors() const { return vertex_colors_; }
  const std::vector<Eigen::Vector3i>& vertex_indices() const { return vertex_indices_; }
  bool set_vertices(const std::vector<Eigen::Vector3f>& v) { vertices_ = v; return true; }
  bool set_vertex_colors(const std::vector<Eigen::Vector3f>& c) { vertex_colors_ = c; return true; }
  bool set_vertex_indices(const std::vector<Eigen::Vector3i>& f) { vertex_indices_ = f; return true; }
  // raw adopt (avoids a second copy of multi-million vertex meshes)
  std::vector<Eigen::Vector3f>* mutable_vertices() { return &vertices_; }
  std::vector<Eigen::Vector3i>* mutable_vertex_indices() { return &vertex_indices_; }
  bool WritePly(const std::string& ply_path) const;        // ASCII, reference format
  bool WritePlyBinary(const std::string& ply_path) const;  // binary_little_endian, for large meshes

 private:
  std::vector<Eigen::Vector3f> vertices_;
  std::vector<Eigen::Vector3f> vertex_colors_;
  std::vector<Eigen::Vector3i> vertex_indices_;
};

}  // namespace vacancy
