// SoftRasterizer.cpp — host layer above the C ABI (see include/SoftRasterizer.hpp).
// Citations are into Liupeter01/Software-Rasterizer.
#include "SoftRasterizer.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

#include "../../../include/srz.h"

namespace SoftRasterizer {

namespace detail {
void load_image_bgr(const std::string &path, std::vector<uint8_t> &bgr, int &W, int &H);

// stand-in for spdlog::error / warn (spdlog is an absent submodule of the reference)
static void log(const char *level, const char *fmt, ...) {
  std::fprintf(stderr, "[%s] ", level);
  va_list ap;
  va_start(ap, fmt);
  std::vfprintf(stderr, fmt, ap);
  va_end(ap);
  std::fputc('\n', stderr);
}
} // namespace detail
using detail::log;

// ---- Shader statics (src/Shader.cpp:7-12) -------------------------------------------------------------------------
glm::vec3 Shader::ka = glm::vec3(0.005f, 0.005f, 0.005f);
glm::vec3 Shader::ks = glm::vec3(0.7937, 0.7937, 0.7937);
float Shader::p = 150;
float Shader::kh = 0.2;
float Shader::kn = 0.1;

// ---- TextureLoader (src/TextureLoader.cpp:3-12) -------------------------------------------------------------------
TextureLoader::TextureLoader(const std::string &path) : m_path(path) {
  int w = 0, h = 0;
  detail::load_image_bgr(path, m_bgr, w, h); // throws "Cannot open file: <path>"
  m_width = (std::size_t)w, m_height = (std::size_t)h;
}
TextureLoader::TextureLoader(const uint8_t *bgr, int width, int height) : m_path("<memory>") {
  if (!bgr || width <= 0 || height <= 0) throw std::runtime_error("Cannot open file: <memory>");
  m_bgr.assign(bgr, bgr + (size_t)width * height * 3);
  m_width = (std::size_t)width, m_height = (std::size_t)height;
}

// ---- Shader (src/Shader.cpp:19-23,94-108) -------------------------------------------------------------------------
Shader::Shader(const std::string &path) : Shader(std::make_shared<TextureLoader>(path)) {}
Shader::Shader(std::shared_ptr<TextureLoader> loader) : texture(std::move(loader)) {}
bool Shader::setFragmentShader(SHADERS_TYPE type) {
  if (static_cast<std::uint8_t>(type) >= 5) {
    log("error", "Set FramentShader Error Due To Invalid Shader Type Input!");
    return false;
  }
  m_type = type;
  return true;
}

// ---- Object::updateModelMatrix (src/Object.cpp:23-31) -------------------------------------------------------------
static glm::mat4 make_model(const glm::vec3 &axis, float angle, const glm::vec3 &translation, const glm::vec3 &scale) {
  auto T = glm::translate(glm::mat4(1.0f), translation);
  auto R = glm::rotate(glm::mat4(1.0f), glm::radians(angle), axis);
  auto S = glm::scale(glm::mat4(1.0f), scale);
  return T * R * S;
}
void Object::updateModelMatrix(const glm::vec3 &axis, float angle, const glm::vec3 &translation, const glm::vec3 &scale) {
  modelMatrix = make_model(axis, angle, translation, scale);
}

// ---- ObjLoader (src/ObjLoader.cpp) --------------------------------------------------------------------------------
ObjLoader::ObjLoader(const std::string &path, const std::string &meshName, const glm::mat4x4 &model)
    : m_path(path), m_meshName(meshName), m_model(model) {}
ObjLoader::ObjLoader(const std::string &path, const std::string &meshName, const glm::vec3 &axis, float angle,
                     const glm::vec3 &translation, const glm::vec3 &scale)
    : ObjLoader(path, meshName) {
  updateModelMatrix(axis, angle, translation, scale);
}
void ObjLoader::updateModelMatrix(const glm::vec3 &axis, float angle, const glm::vec3 &translation, const glm::vec3 &scale) {
  m_model = make_model(axis, angle, translation, scale);
}

namespace {
struct VKey { // equality semantics of Vertex::operator== (float ==, so -0 == +0)
  float v[11];
  bool operator==(const VKey &o) const {
    for (int i = 0; i < 11; ++i)
      if (!(v[i] == o.v[i])) return false;
    return true;
  }
};
struct VKeyHash {
  size_t operator()(const VKey &k) const {
    size_t seed = 0;
    for (int i = 0; i < 11; ++i) {
      float f = k.v[i] == 0.0f ? 0.0f : k.v[i]; // -0 and +0 hash alike
      uint32_t u;
      std::memcpy(&u, &f, 4);
      seed ^= (size_t)u + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    }
    return seed;
  }
};

// Tools::calculateNormalWithWeight (src/Tools.cpp:234-248)
glm::vec3 normal_with_weight(const glm::vec3 &pa, const glm::vec3 &pb, const glm::vec3 &pc) {
  const glm::vec3 AB = pb - pa, AC = pc - pa;
  glm::vec3 normal = glm::cross(AB, AC);
  const float length = glm::length(normal);
  const float arc_sin_degree = length / (glm::length(AB) * glm::length(AC));
  if (!(-(1e-8) <= length && length <= 1e-8)) normal = normal * (glm::asin(arc_sin_degree) / length);
  return glm::normalize(normal);
}

int fix_index(int i, int n) { return i > 0 ? i - 1 : n + i; }
} // namespace

// tinyobj::LoadObj (triangulate = true) + processingVertexData (src/ObjLoader.cpp:78-233).
// tinyobjloader is an absent submodule; its documented behaviour is restated: 1-based / negative-relative indices,
// fan triangulation, missing colours = 1, absent vt / vn → no texcoord / normal.
std::optional<std::unique_ptr<Mesh>> ObjLoader::startLoadingFromFile(const std::string &objName) {
  std::ifstream in(m_path);
  if (!in) {
    log("error", "[TinyObjReader]: Error Occured! Cannot open file [%s]", m_path.c_str());
    throw std::runtime_error("LoadObj Error");
  }
  std::vector<glm::vec3> pos, col, nrm;
  std::vector<glm::vec2> tex;
  struct Corner {
    int v, t, n;
  };
  std::vector<Corner> corners;
  std::string line, shape_name;
  while (std::getline(in, line)) {
    const char *s = line.c_str();
    while (*s == ' ' || *s == '\t') ++s;
    if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
      float v[6] = {0, 0, 0, 1, 1, 1};
      int n = std::sscanf(s + 2, "%f %f %f %f %f %f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]);
      pos.emplace_back(v[0], v[1], v[2]);
      col.push_back(n >= 6 ? glm::vec3(v[3], v[4], v[5]) : glm::vec3(1.0f));
    } else if (s[0] == 'v' && s[1] == 't') {
      float u = 0, v = 0;
      std::sscanf(s + 3, "%f %f", &u, &v);
      tex.emplace_back(u, v);
    } else if (s[0] == 'v' && s[1] == 'n') {
      float x = 0, y = 0, z = 0;
      std::sscanf(s + 3, "%f %f %f", &x, &y, &z);
      nrm.emplace_back(x, y, z);
    } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
      std::vector<Corner> poly;
      std::istringstream ss(s + 2);
      std::string tok;
      while (ss >> tok) {
        Corner c{0, -1, -1};
        int vi = 0, ti = 0, ni = 0;
        if (std::sscanf(tok.c_str(), "%d/%d/%d", &vi, &ti, &ni) == 3) {
          c.t = fix_index(ti, (int)tex.size()), c.n = fix_index(ni, (int)nrm.size());
        } else if (std::sscanf(tok.c_str(), "%d//%d", &vi, &ni) == 2) {
          c.n = fix_index(ni, (int)nrm.size());
        } else if (std::sscanf(tok.c_str(), "%d/%d", &vi, &ti) == 2) {
          c.t = fix_index(ti, (int)tex.size());
        } else if (std::sscanf(tok.c_str(), "%d", &vi) != 1) {
          continue;
        }
        c.v = fix_index(vi, (int)pos.size());
        poly.push_back(c);
      }
      for (size_t k = 2; k < poly.size(); ++k) corners.push_back(poly[0]), corners.push_back(poly[k - 1]), corners.push_back(poly[k]);
    } else if ((s[0] == 'o' || s[0] == 'g') && (s[1] == ' ' || s[1] == '\t')) {
      shape_name = s + 2;
    }
  }

  bool noNormal = true;
  std::vector<Vertex> vertices;
  std::vector<uint32_t> indices;
  std::unordered_map<VKey, uint32_t, VKeyHash> unique;
  for (const Corner &c : corners) {
    if (c.v < 0 || c.v >= (int)pos.size()) {
      log("error", "[TinyObjReader]: Error Occured! vertex index out of range");
      throw std::runtime_error("LoadObj Error");
    }
    Vertex vertex;
    vertex.position = pos[c.v];
    vertex.color = col[c.v];
    if (c.n >= 0 && c.n < (int)nrm.size()) {
      noNormal = false;
      vertex.normal = glm::normalize(nrm[c.n]); // (:141-145)
    }
    if (c.t >= 0 && c.t < (int)tex.size()) vertex.texCoord = glm::vec2(tex[c.t].x, 1.0f - tex[c.t].y); // (:150-152)
    VKey k{{vertex.position.x, vertex.position.y, vertex.position.z, vertex.color.x, vertex.color.y, vertex.color.z,
            vertex.normal.x, vertex.normal.y, vertex.normal.z, vertex.texCoord.x, vertex.texCoord.y}};
    auto it = unique.find(k);
    if (it == unique.end()) {
      it = unique.emplace(k, (uint32_t)vertices.size()).first;
      vertices.push_back(vertex);
    }
    indices.push_back(it->second);
  }
  std::vector<glm::uvec3> faces(indices.size() / 3);
  for (size_t i = 0; i < indices.size() / 3; ++i) {
    uint32_t a = indices[3 * i], b = indices[3 * i + 1], c = indices[3 * i + 2];
    faces[i] = glm::uvec3(a, b, c);
    if (noNormal) { // (:181-188)
      Vertex &A = vertices[a], &B = vertices[b], &C = vertices[c];
      A.normal = normal_with_weight(A.position, B.position, C.position);
      B.normal = normal_with_weight(B.position, C.position, A.position);
      C.normal = normal_with_weight(C.position, A.position, B.position);
    }
  }
  auto mesh = std::make_unique<Mesh>(objName.empty() ? shape_name : objName, std::move(vertices), std::move(faces));
  return mesh;
}

// ---- Scene --------------------------------------------------------------------------------------------------------
Scene::Scene(const std::string &sceneName, const glm::vec3 &eye, const glm::vec3 &center, const glm::vec3 &up,
             glm::vec3, std::size_t, float)
    : m_sceneName(sceneName), m_eye(eye), m_center(center), m_up(up) {
  setViewMatrix(eye, center, up);
}

bool Scene::addGraphicObj(const std::string &path, const std::string &meshName, const glm::vec3 &axis, float angle,
                          const glm::vec3 &translation, const glm::vec3 &scale_) {
  if (m_loadedObjs.find(meshName) != m_loadedObjs.end()) {
    log("error", "Add Graphic Obj Error! This Object has already been identified");
    return false;
  }
  m_loadedObjs[meshName].loader = std::make_unique<ObjLoader>(path, meshName, axis, angle, translation, scale_);
  m_objOrder.push_back(meshName);
  return true;
}
bool Scene::addGraphicObj(const std::string &path, const std::string &meshName) {
  if (m_loadedObjs.find(meshName) != m_loadedObjs.end()) {
    log("error", "This Object has already been identified");
    return false;
  }
  m_loadedObjs[meshName].loader = std::make_unique<ObjLoader>(path, meshName);
  m_objOrder.push_back(meshName);
  return true;
}
bool Scene::addGraphicObj(std::unique_ptr<Object> object, const std::string &objectName) {
  if (m_loadedObjs.find(objectName) != m_loadedObjs.end()) {
    log("error", "This Object has already been identified");
    return false;
  }
  m_loadedObjs[objectName].loader = std::nullopt;
  m_loadedObjs[objectName].mesh = std::move(object);
  m_objOrder.push_back(objectName);
  return true;
}

bool Scene::startLoadingMesh(const std::string &meshName) {
  auto it = m_loadedObjs.find(meshName);
  if (it == m_loadedObjs.end()) {
    log("error", "Start Loading Mesh Failed! Because There is nothing found in m_loadedObjs");
    return false;
  }
  if (it->second.mesh != nullptr) {
    log("error", "Start Loading Mesh Failed! Because %s Has Already Loaded into m_loadedObjs", meshName.c_str());
    return false;
  }
  try {
    if (!it->second.loader.has_value()) return false;
    ObjLoader &ld = *it->second.loader.value();
    auto mesh_op = ld.startLoadingFromFile(meshName);
    if (!mesh_op.has_value()) {
      log("error", "Start Loading Mesh Failed! Because Loading Internel Error!");
      return false;
    }
    it->second.mesh = std::move(mesh_op.value());
  } catch (const std::exception &e) {
    log("error", "Start Loading Mesh Failed! Reason: %s", e.what());
    return false;
  }
  return true;
}

std::optional<std::shared_ptr<Object>> Scene::getMeshObj(const std::string &meshName) {
  auto it = m_loadedObjs.find(meshName);
  if (it == m_loadedObjs.end()) {
    log("error", "Get Mesh Failed! Because There is nothing found in m_loadedObjs");
    return std::nullopt;
  }
  if (it->second.mesh == nullptr) {
    log("error", "You Have to get Mesh Object After Deploy startLoadingMesh");
    return std::nullopt;
  }
  return std::shared_ptr<Object>(it->second.mesh.get(), [](Object *) {});
}

bool Scene::addShader(const std::string &shaderName, const std::string &texturePath, SHADERS_TYPE type) {
  if (m_shaders.find(shaderName) != m_shaders.end()) {
    log("error", "Add Shader Failed! Because Shader %s Already Exist!", shaderName.c_str());
    return false;
  }
  try {
    auto sh = std::make_shared<Shader>(texturePath);
    sh->setFragmentShader(type);
    m_shaders[shaderName] = sh;
  } catch (const std::exception &e) {
    log("error", "Add Shader Failed! Reason: %s", e.what());
    return false;
  }
  return true;
}
bool Scene::addShader(const std::string &shaderName, std::shared_ptr<TextureLoader> text, SHADERS_TYPE type) {
  if (m_shaders.find(shaderName) != m_shaders.end()) {
    log("error", "Add Shader Failed! Because Shader %s Already Exist!", shaderName.c_str());
    return false;
  }
  auto sh = std::make_shared<Shader>(std::move(text));
  sh->setFragmentShader(type);
  m_shaders[shaderName] = sh;
  return true;
}

bool Scene::bindShader2Mesh(const std::string &meshName, const std::string &shaderName) {
  auto it = m_loadedObjs.find(meshName);
  if (it == m_loadedObjs.end()) {
    log("error", "Bind Shader To Mesh Failed! Because Loaded Mesh %s Not found!", meshName.c_str());
    return false;
  }
  auto sh = m_shaders.find(shaderName);
  if (sh == m_shaders.end()) {
    log("error", "Bind Shader To Mesh Failed! Because Shader %s Not found!", shaderName.c_str());
    return false;
  }
  if (!it->second.mesh) { // the reference dereferences a null mesh here; we report instead
    log("error", "Bind Shader To Mesh Failed! Because Mesh %s Is Not Loaded Yet!", meshName.c_str());
    return false;
  }
  it->second.mesh->bindShader2Mesh(sh->second);
  return true;
}

void Scene::addLight(std::string name, std::shared_ptr<light_struct> light) {
  for (auto &kv : m_lights)
    if (kv.first == name) {
      log("warn", "Add Light Success! Because Light %s Already Been Added!", name.c_str());
      return;
    }
  m_lights.emplace_back(std::move(name), std::move(light));
}
void Scene::addLights(std::vector<std::pair<std::string, std::shared_ptr<light_struct>>> lights) {
  for (auto &kv : lights) addLight(kv.first, kv.second);
}

bool Scene::setModelMatrix(const std::string &meshName, const glm::vec3 &axis, float angle, const glm::vec3 &translation,
                           const glm::vec3 &scale_) {
  auto it = m_loadedObjs.find(meshName);
  if (it == m_loadedObjs.end() || !it->second.mesh) {
    log("error", "Editing Model Matrix Failed! Because %s Not Found", meshName.c_str());
    return false;
  }
  it->second.mesh->updateModelMatrix(axis, angle, translation, scale_);
  return true;
}

// (src/Scene.cpp:263-271)
void Scene::setViewMatrix(const glm::vec3 &eye, const glm::vec3 &center, const glm::vec3 &up) {
  m_eye = eye, m_center = center, m_up = up;
  m_view = glm::lookAtLH(eye, center, up);
}
// (src/Scene.cpp:273-294) — fovy is handed to a RADIANS api as-is, exactly like the reference
void Scene::setProjectionMatrix(float fovy, float zNear, float zFar) {
  m_fovy = fovy, m_near = zNear, m_far = zFar;
  scale = (m_far - m_near) / 2.0f;
  offset = (m_far + m_near) / 2.0f;
  m_projection = glm::perspectiveLH_NO(fovy, m_aspectRatio, zNear, zFar);
}
// (src/Scene.cpp:314-335)
void Scene::setNDCMatrix(std::size_t width, std::size_t height) {
  m_width = width, m_height = height;
  if (!m_height) throw std::runtime_error("Height cannot be zero!");
  m_aspectRatio = static_cast<float>(m_width) / static_cast<float>(m_height);
  glm::mat4 matrix(1.0f);
  matrix[0][0] = width / 2.0f * m_aspectRatio;
  matrix[1][1] = height / 2.0f;
  matrix[3][0] = width / 2.0f;
  matrix[3][1] = height / 2.0f;
  m_ndcToScreenMatrix = matrix;
}

// (src/Scene.cpp:296-312)
std::vector<light_struct> Scene::loadLights() {
  if (reference_exact_lights) return std::vector<light_struct>(m_lights.size()); // as written: default lights
  std::vector<light_struct> res;
  for (auto &kv : m_lights) res.push_back(*kv.second);
  return res;
}

static inline glm::vec3 to_vec3(const glm::vec4 &v) { return glm::vec3(v.x / v.w, v.y / v.w, v.z / v.w); } // src/Tools.cpp:74-76

// (src/Scene.cpp:903-964)
std::vector<Scene::ObjTuple> Scene::loadTriangleStream() {
  std::vector<ObjTuple> stream;
  for (const std::string &name : m_objOrder) {
    const ObjInfo &obj = m_loadedObjs[name];
    if (!obj.mesh) continue;
    const auto &mesh = obj.mesh;
    const glm::mat4 &modelMatrix = mesh->getModelMatrix();
    const glm::mat4 NDC_MVP = m_ndcToScreenMatrix * m_projection * m_view * modelMatrix;
    const glm::mat4 Normal_M = glm::transpose(glm::inverse(modelMatrix));
    const auto &faces = mesh->getFaces();
    const auto &verts = mesh->getVertices();
    std::vector<RasterTriangle> ret(faces.size());
    for (size_t fi = 0; fi < faces.size(); ++fi) {
      const glm::uvec3 &face = faces[fi];
      const unsigned idx[3] = {face.x, face.y, face.z};
      for (int k = 0; k < 3; ++k) {
        const Vertex &V = verts[idx[k]];
        glm::vec3 P = to_vec3(NDC_MVP * glm::vec4(V.position, 1.0f));
        P.z = P.z * scale + offset; // Z-Depth
        glm::vec3 N = to_vec3(Normal_M * glm::vec4(V.normal, 1.0f));
        ret[fi].pos[k][0] = P.x, ret[fi].pos[k][1] = P.y, ret[fi].pos[k][2] = P.z;
        ret[fi].nrm[k][0] = N.x, ret[fi].nrm[k][1] = N.y, ret[fi].nrm[k][2] = N.z;
        ret[fi].uv[k][0] = V.texCoord.x, ret[fi].uv[k][1] = V.texCoord.y;
      }
    }
    stream.emplace_back(mesh->shader(), std::move(ret));
  }
  return stream;
}

std::vector<Scene::MeshDraw> Scene::meshDraws() {
  std::vector<MeshDraw> out;
  for (const std::string &name : m_objOrder) {
    const ObjInfo &obj = m_loadedObjs[name];
    if (!obj.mesh) continue;
    const glm::mat4 &modelMatrix = obj.mesh->getModelMatrix();
    MeshDraw d;
    d.name = name, d.mesh = obj.mesh.get(), d.shader = obj.mesh->shader();
    d.ndc_mvp = m_ndcToScreenMatrix * m_projection * m_view * modelMatrix;
    d.normal_m = glm::transpose(glm::inverse(modelMatrix));
    out.push_back(std::move(d));
  }
  return out;
}

// ---- RenderingPipeline (src/Render.cpp) ---------------------------------------------------------------------------
RenderingPipeline::RenderingPipeline() : RenderingPipeline(800, 600) {}
RenderingPipeline::RenderingPipeline(std::size_t width, std::size_t height) : m_width(width), m_height(height) {
  for (auto &c : m_channels) c.resize(width * height);
  m_zBuffer.resize(width * height);
  m_frameBuffer8.resize(width * height * 3);
  clear(Buffers::Color | Buffers::Depth);
}
RenderingPipeline::~RenderingPipeline() {}
void RenderingPipeline::clearFrameBuffer() {
  for (auto &c : m_channels) std::fill(c.begin(), c.end(), 0.0f);
}
void RenderingPipeline::clearZDepth() { std::fill(m_zBuffer.begin(), m_zBuffer.end(), std::numeric_limits<float>::infinity()); }
void RenderingPipeline::clear(Buffers flags) {
  const bool c = (flags & Buffers::Color) == Buffers::Color, d = (flags & Buffers::Depth) == Buffers::Depth;
  if (m_target) { // planes are in HBM: clear there (free when a draw follows immediately)
    if (srz_target_clear(m_ctx, m_target, c ? 1 : 0, d ? 1 : 0) != SRZ_OK) throw std::runtime_error(std::string("clear: ") + srz_last_error(m_ctx));
    m_hostStale = m_hostStale || c || d;
    return;
  }
  if (c) clearFrameBuffer();
  if (d) clearZDepth();
  m_justCleared = c && d;
}
void RenderingPipeline::syncToHost() {
  if (!m_target || !m_hostStale) return;
  if (srz_target_read(m_ctx, m_target, m_zBuffer.data(), m_channels[0].data(), m_channels[1].data(), m_channels[2].data()) != SRZ_OK)
    throw std::runtime_error(std::string("framebuffer read-back: ") + srz_last_error(m_ctx));
  m_hostStale = false;
}
bool RenderingPipeline::addScene(std::shared_ptr<Scene> scene, std::optional<std::string> name) {
  try {
    if (scene == nullptr) return false;
    if (name.has_value()) scene->m_sceneName = name.value();
    scene->setNDCMatrix(m_width, m_height);
    for (auto &kv : m_scenes)
      if (kv.first == scene->m_sceneName) {
        log("error", "Add Scene Failed! Scene Already Exist");
        return false;
      }
    m_scenes.emplace_back(scene->m_sceneName, scene);
  } catch (const std::exception &e) {
    log("error", "Add Scene Failed! Reason: %s", e.what());
    return false;
  }
  return true;
}
// (src/Render.cpp:57-64): draw, cv::merge, convertTo(CV_8UC3) = saturate_cast<uchar>(cvRound(v)); no imshow here
void RenderingPipeline::display(Primitive type) {
  draw(type);
  if (m_target) { // resolve on the device, 3 bytes per pixel come back
    if (srz_target_read_bgr8(m_ctx, m_target, m_frameBuffer8.data()) != SRZ_OK)
      throw std::runtime_error(std::string("display: ") + srz_last_error(m_ctx));
    return;
  }
  const size_t n = m_width * m_height;
  for (size_t i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) {
      float v = m_channels[c][i];
      long r = (v == v && v > -1e9f && v < 1e9f) ? std::lrintf(v) : (v > 0 ? 255 : 0);
      m_frameBuffer8[i * 3 + c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

// ---- TraditionalRasterizer (src/Rasterizer.cpp:183-240) -----------------------------------------------------------
TraditionalRasterizer::TraditionalRasterizer() : RenderingPipeline() { init(); }
TraditionalRasterizer::TraditionalRasterizer(std::size_t width, std::size_t height) : RenderingPipeline(width, height) { init(); }
void TraditionalRasterizer::init() {
  // the host layer is compiled against include/srz.h: a libsrz.so of another ABI version (stale build, LD_LIBRARY_PATH) is refused
  if (srz_abi_version() != SRZ_ABI_VERSION)
    throw std::runtime_error("TraditionalRasterizer: libsrz.so reports SRZ_ABI_VERSION " + std::to_string(srz_abi_version()) +
                             ", libsrz_host.so was built against " + std::to_string(SRZ_ABI_VERSION) + " — rebuild both");
  int dev = 0;
  if (const char *e = std::getenv("SRZ_DEVICE")) dev = std::atoi(e);
  int rc = srz_create(&m_ctx, dev);
  if (rc != SRZ_OK) throw std::runtime_error(std::string("TraditionalRasterizer: ") + srz_last_error(nullptr));
  rc = srz_target_create(m_ctx, (int)m_width, (int)m_height, &m_target);
  if (rc != SRZ_OK) {
    std::string msg = srz_last_error(m_ctx);
    srz_destroy(m_ctx);
    m_ctx = nullptr;
    throw std::runtime_error("TraditionalRasterizer: " + msg);
  }
  m_hostStale = false; // the constructor's clear() already filled the host planes with the same values
}
TraditionalRasterizer::~TraditionalRasterizer() {
  for (auto &kv : m_sceneSets) srz_frameset_destroy(m_ctx, kv.second);
  srz_target_destroy(m_ctx, m_target);
  srz_destroy(m_ctx);
}

int TraditionalRasterizer::textureSlot(const std::shared_ptr<Shader> &sh) {
  TextureLoader *tl = sh->getTextureObject().get();
  auto it = m_texSlots.find(tl);
  if (it == m_texSlots.end()) {
    int slot = (int)m_texSlots.size();
    if (slot >= 64) throw std::runtime_error("draw: more than 64 distinct textures");
    int rc = srz_texture_upload(m_ctx, slot, tl->bgr().data(), (int)tl->width(), (int)tl->height(), (int)tl->width() * 3);
    if (rc != SRZ_OK) throw std::runtime_error(std::string("draw: ") + srz_last_error(m_ctx));
    it = m_texSlots.emplace(tl, slot).first;
    m_texOwners.push_back(sh->getTextureObject()); // keep the key's identity stable
  }
  return it->second;
}

void RenderingPipeline::finish() {
  if (m_ctx && srz_sync(m_ctx) != SRZ_OK) throw std::runtime_error(std::string("finish: ") + srz_last_error(m_ctx));
}

void TraditionalRasterizer::draw(Primitive type) {
  if ((type != Primitive::LINES) && (type != Primitive::TRIANGLES)) {
    log("error", "Primitive Type is not supported!");
    throw std::runtime_error("Primitive Type is not supported!");
  }
  const int prim = type == Primitive::LINES ? SRZ_PRIMITIVE_LINES : SRZ_PRIMITIVE_TRIANGLES;
  last_stats = Stats();
  for (auto &kv : m_scenes) {
    Scene &scene = *kv.second;
    if (device_vertex_stage) { // meshes resident on the GPU, vertex stage there (k_vertex)
      std::vector<light_struct> lights = scene.loadLights();
      const glm::vec3 eye = scene.loadEyeVec();
      std::vector<srz_light> L(lights.size());
      for (size_t i = 0; i < lights.size(); ++i) {
        L[i].pos[0] = lights[i].position.x, L[i].pos[1] = lights[i].position.y, L[i].pos[2] = lights[i].position.z;
        L[i].intensity[0] = lights[i].intensity.x, L[i].intensity[1] = lights[i].intensity.y, L[i].intensity[2] = lights[i].intensity.z;
      }
      std::vector<srz_mesh_draw> D;
      for (Scene::MeshDraw &md : scene.meshDraws()) {
        const auto &faces = md.mesh->getFaces();
        if (faces.empty()) continue;
        if (!md.shader) throw std::runtime_error("draw: a mesh with triangles has no shader bound (bindShader2Mesh)"); // D14
        // The GPU copy of a mesh is reused only while the mesh is PROVEN unchanged: the reference re-reads vertices and
        // faces on every draw() (src/Scene.cpp:927-947) and both are public members, so a content hash (≈13 µs for spot)
        // is the only proof there is — an in-place edit, or a new mesh at a recycled address, re-uploads.
        const auto &V = md.mesh->getVertices();
        // (four independent multiply-xorshift lanes over 32-byte blocks: the multiply chain of a single lane is what a
        // one-lane version of this hash waits for — 50 µs for spot's 212 KB against 13 µs)
        auto hash_bytes = [](const void *p, size_t n, uint64_t h) {
          const unsigned char *b = static_cast<const unsigned char *>(p);
          uint64_t h0 = h, h1 = h ^ 0x9E3779B97F4A7C15ull, h2 = h + 0xD1B54A32D192ED03ull, h3 = ~h;
          for (; n >= 32; n -= 32, b += 32) {
            uint64_t w[4];
            std::memcpy(w, b, 32);
            h0 = (h0 ^ w[0]) * 0x9E3779B97F4A7C15ull, h0 ^= h0 >> 29;
            h1 = (h1 ^ w[1]) * 0xC2B2AE3D27D4EB4Full, h1 ^= h1 >> 31;
            h2 = (h2 ^ w[2]) * 0x165667B19E3779F9ull, h2 ^= h2 >> 27;
            h3 = (h3 ^ w[3]) * 0xD6E8FEB86659FD93ull, h3 ^= h3 >> 32;
          }
          h = (h0 ^ (h1 << 1 | h1 >> 63)) * 0x9E3779B97F4A7C15ull;
          h = (h ^ h2 ^ (h3 << 7 | h3 >> 57)) * 0xC2B2AE3D27D4EB4Full, h ^= h >> 29;
          for (; n >= 8; n -= 8, b += 8) {
            uint64_t w;
            std::memcpy(&w, b, 8);
            h = (h ^ w) * 0x9E3779B97F4A7C15ull, h ^= h >> 29;
          }
          for (; n; --n, ++b) h = (h ^ *b) * 0x100000001B3ull;
          return h;
        };
        uint64_t hash = hash_bytes(V.data(), V.size() * sizeof(V[0]), 0xcbf29ce484222325ull ^ V.size());
        hash = hash_bytes(faces.data(), faces.size() * sizeof(faces[0]), hash ^ (faces.size() << 1));
        auto it = m_meshSlots.find(md.mesh);
        if (it == m_meshSlots.end() || it->second.hash != hash || it->second.n_faces != faces.size()) {
          const int slot = it == m_meshSlots.end() ? (int)m_meshSlots.size() : it->second.slot;
          if (slot >= 256) throw std::runtime_error("draw: more than 256 meshes");
          std::vector<srz_vertex> v(V.size());
          for (size_t i = 0; i < V.size(); ++i) {
            v[i].pos[0] = V[i].position.x, v[i].pos[1] = V[i].position.y, v[i].pos[2] = V[i].position.z;
            v[i].nrm[0] = V[i].normal.x, v[i].nrm[1] = V[i].normal.y, v[i].nrm[2] = V[i].normal.z;
            v[i].uv[0] = V[i].texCoord.x, v[i].uv[1] = V[i].texCoord.y;
          }
          std::vector<uint32_t> f(faces.size() * 3);
          for (size_t i = 0; i < faces.size(); ++i) f[3 * i] = faces[i].x, f[3 * i + 1] = faces[i].y, f[3 * i + 2] = faces[i].z;
          int rc = srz_mesh_upload(m_ctx, slot, v.data(), (uint32_t)v.size(), f.data(), (uint32_t)faces.size());
          if (rc != SRZ_OK) throw std::runtime_error(std::string("draw: ") + srz_last_error(m_ctx));
          m_meshSlots[md.mesh] = MeshSlot{slot, faces.size(), hash};
          it = m_meshSlots.find(md.mesh);
        }
        srz_mesh_draw d{};
        d.mesh_id = it->second.slot, d.shader = (int)md.shader->type(), d.tex_id = -1;
        const SHADERS_TYPE st = md.shader->type();
        if (st == SHADERS_TYPE::TEXTURE || st == SHADERS_TYPE::DISPLACEMENT || st == SHADERS_TYPE::BUMP) d.tex_id = textureSlot(md.shader);
        std::memcpy(d.ndc_mvp, md.ndc_mvp.data(), 64), std::memcpy(d.normal_m, md.normal_m.data(), 64);
        D.push_back(d);
      }
      srz_scene_frame sf{};
      sf.width = (int)m_width, sf.height = (int)m_height;
      sf.eye[0] = eye.x, sf.eye[1] = eye.y, sf.eye[2] = eye.z;
      sf.ka[0] = Shader::ka.x, sf.ka[1] = Shader::ka.y, sf.ka[2] = Shader::ka.z;
      sf.ks[0] = Shader::ks.x, sf.ks[1] = Shader::ks.y, sf.ks[2] = Shader::ks.z;
      sf.p = Shader::p, sf.kh = Shader::kh, sf.kn = Shader::kn;
      sf.zscale = scene.depthScale(), sf.zoffset = scene.depthOffset();
      sf.n_lights = (uint32_t)L.size(), sf.lights = L.data();
      sf.n_draws = (uint32_t)D.size(), sf.draws = D.data();
      sf.flags = SRZ_EXACT_SPLIT; // whether the target was just cleared is the target's own state
      srz_stats st{};
      srz_frameset *&set = m_sceneSets[&scene];
      if (set && srz_sceneset_update(m_ctx, set, &sf, 1) != SRZ_OK) { // structure changed: rebuild
        srz_frameset_destroy(m_ctx, set);
        set = nullptr;
      }
      if (!set && srz_sceneset_create(m_ctx, &sf, 1, &set) != SRZ_OK) throw std::runtime_error(std::string("draw: ") + srz_last_error(m_ctx));
      int rc = srz_target_draw(m_ctx, m_target, prim, set, collect_stats ? &st : nullptr);
      if (rc != SRZ_OK) throw std::runtime_error(std::string("draw: ") + srz_last_error(m_ctx));
      m_hostStale = true;
      last_stats.n_tris += st.n_tris, last_stats.n_culled += st.n_culled, last_stats.pixel_tests += st.pixel_tests;
      last_stats.fragments += st.fragments, last_stats.shaded += st.shaded, last_stats.visible += st.visible;
      last_stats.visible_textured += st.visible_textured;
      continue;
    }
    std::vector<Scene::ObjTuple> stream = scene.loadTriangleStream();
    std::vector<light_struct> lights = scene.loadLights();
    const glm::vec3 eye = scene.loadEyeVec();

    std::vector<srz_light> L(lights.size());
    for (size_t i = 0; i < lights.size(); ++i) {
      L[i].pos[0] = lights[i].position.x, L[i].pos[1] = lights[i].position.y, L[i].pos[2] = lights[i].position.z;
      L[i].intensity[0] = lights[i].intensity.x, L[i].intensity[1] = lights[i].intensity.y, L[i].intensity[2] = lights[i].intensity.z;
    }
    std::vector<srz_batch> B;
    for (auto &tup : stream) {
      const std::shared_ptr<Shader> &sh = std::get<0>(tup);
      const std::vector<RasterTriangle> &tris = std::get<1>(tup);
      if (tris.empty()) continue;
      if (!sh) throw std::runtime_error("draw: a mesh with triangles has no shader bound (bindShader2Mesh)"); // D14
      srz_batch b{};
      b.shader = (int)sh->type();
      b.tex_id = -1;
      b.n_tris = (uint32_t)tris.size();
      b.tris = reinterpret_cast<const srz_tri *>(tris.data());
      const bool needs_tex = sh->type() == SHADERS_TYPE::TEXTURE || sh->type() == SHADERS_TYPE::DISPLACEMENT || sh->type() == SHADERS_TYPE::BUMP;
      if (needs_tex) b.tex_id = textureSlot(sh);
      B.push_back(b);
    }
    srz_frame fr{};
    fr.width = (int)m_width, fr.height = (int)m_height;
    fr.eye[0] = eye.x, fr.eye[1] = eye.y, fr.eye[2] = eye.z;
    fr.ka[0] = Shader::ka.x, fr.ka[1] = Shader::ka.y, fr.ka[2] = Shader::ka.z;
    fr.ks[0] = Shader::ks.x, fr.ks[1] = Shader::ks.y, fr.ks[2] = Shader::ks.z;
    fr.p = Shader::p, fr.kh = Shader::kh, fr.kn = Shader::kn;
    fr.n_lights = (uint32_t)L.size(), fr.lights = L.data();
    fr.n_batches = (uint32_t)B.size(), fr.batches = B.data();
    fr.flags = SRZ_EXACT_SPLIT;
    srz_stats st{};
    srz_frameset *tmp = nullptr;
    if (srz_frameset_create(m_ctx, &fr, 1, &tmp) != SRZ_OK) throw std::runtime_error(std::string("draw: ") + srz_last_error(m_ctx));
    int rc = srz_target_draw(m_ctx, m_target, prim, tmp, collect_stats ? &st : nullptr);
    srz_frameset_destroy(m_ctx, tmp);
    if (rc != SRZ_OK) throw std::runtime_error(std::string("draw: ") + srz_last_error(m_ctx));
    m_hostStale = true;
    last_stats.n_tris += st.n_tris, last_stats.n_culled += st.n_culled, last_stats.pixel_tests += st.pixel_tests;
    last_stats.fragments += st.fragments, last_stats.shaded += st.shaded, last_stats.visible += st.visible;
    last_stats.visible_textured += st.visible_textured;
  }
}

} // namespace SoftRasterizer
