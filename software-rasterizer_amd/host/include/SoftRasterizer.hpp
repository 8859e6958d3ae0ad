// SoftRasterizer.hpp — host-side mirror of the reference's C++ API for the raster path ONLY:
//   SoftRasterizer::TraditionalRasterizer / RenderingPipeline   include/render/Rasterizer.hpp, include/base/Render.hpp
//   SoftRasterizer::Scene                                        include/scene/Scene.hpp (raster-related members)
//   SoftRasterizer::Shader / SHADERS_TYPE / TextureLoader        include/shader/Shader.hpp, include/loader/TextureLoader.hpp
//   SoftRasterizer::light_struct, Vertex, Object, Mesh, ObjLoader
// Same names, argument meaning and error behaviour (bool + log for Scene/addScene, std::runtime_error from draw()
// and from the TextureLoader / Triangle-style constructors).  draw() does not rasterise on the CPU: it packs what the
// reference's draw() pulls from the scene into an srz_frame and calls the C ABI (include/srz.h) → gfx950 kernels.
// There is no CPU fallback; constructing a TraditionalRasterizer without a usable MI355X throws.
//
// Out of scope: ray / path tracing (RayTracing, PathTracing, BVH, Sphere, Cube, Material BRDFs, lights as
// emissive objects), imshow.  Deviations from the reference are listed in DESIGN.md ("Documented deviations").
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "glm_min.hpp"

struct srz_ctx;
struct srz_target;
struct srz_frameset;

namespace SoftRasterizer {

// ---- include/light/Light.hpp:8-45 -------------------------------------------------------------------------------
struct light_struct {
  light_struct() : position(0.f), intensity(0.f) {}
  light_struct(const glm::vec3 &pos, const glm::vec3 &intense) : position(pos), intensity(intense) {}
  glm::vec3 position;
  glm::vec3 intensity;
};

// ---- include/object/Object.hpp:17-31 ----------------------------------------------------------------------------
struct Vertex {
  Vertex() : position(0.f), normal(0.f), texCoord(0.f), color(1.0f) {}
  Vertex(const glm::vec3 &p, const glm::vec3 &n, const glm::vec2 &t, const glm::vec3 &c = glm::vec3(1.0f))
      : position(p), normal(n), texCoord(t), color(c) {}
  glm::vec3 position, normal;
  glm::vec2 texCoord;
  glm::vec3 color;
  bool operator==(const Vertex &o) const {
    return position == o.position && color == o.color && normal == o.normal && texCoord == o.texCoord;
  }
};

// ---- include/loader/TextureLoader.hpp ---------------------------------------------------------------------------
// Holds what cv::imread(path) returns in the reference: 8-bit BGR, row-major, top row first, alpha dropped.
class TextureLoader {
public:
  explicit TextureLoader(const std::string &path); // throws std::runtime_error("Cannot open file: " + path)
  TextureLoader(const uint8_t *bgr, int width, int height); // from memory (not in the reference; for tests/tools)
  virtual ~TextureLoader() = default;
  const std::vector<uint8_t> &bgr() const { return m_bgr; }
  std::size_t width() const { return m_width; }
  std::size_t height() const { return m_height; }
  const std::string &path() const { return m_path; }

private:
  std::vector<uint8_t> m_bgr;
  std::string m_path;
  std::size_t m_width = 0, m_height = 0;
};

// ---- include/shader/Shader.hpp ----------------------------------------------------------------------------------
enum class SHADERS_TYPE : std::uint8_t { NORMAL = 0, TEXTURE, PHONG, DISPLACEMENT, BUMP };

struct Shader {
  static glm::vec3 ka, ks; // process-wide statics exactly like the reference (src/Shader.cpp:7-12)
  static float p, kh, kn;
  explicit Shader(const std::string &path);
  explicit Shader(std::shared_ptr<TextureLoader> loader);
  std::shared_ptr<TextureLoader> &getTextureObject() { return texture; }
  bool setFragmentShader(SHADERS_TYPE type);
  SHADERS_TYPE type() const { return m_type; }

private:
  std::shared_ptr<TextureLoader> texture;
  SHADERS_TYPE m_type = SHADERS_TYPE::NORMAL;
};

// ---- include/object/Object.hpp, Mesh.hpp (raster-relevant part) -------------------------------------------------
struct Object {
  virtual ~Object() = default;
  virtual const std::vector<Vertex> &getVertices() const = 0;
  virtual const std::vector<glm::uvec3> &getFaces() const = 0;
  void updateModelMatrix(const glm::vec3 &axis, float angle, const glm::vec3 &translation, const glm::vec3 &scale);
  const glm::mat4x4 &getModelMatrix() const { return modelMatrix; }
  virtual void bindShader2Mesh(std::shared_ptr<Shader> shader) { m_shader = std::move(shader); }
  const std::shared_ptr<Shader> &shader() const { return m_shader; }
  glm::mat4x4 modelMatrix = glm::mat4x4(1.0f);

protected:
  std::shared_ptr<Shader> m_shader;
};

struct Mesh : public Object {
  Mesh(std::string name, std::vector<Vertex> &&v, std::vector<glm::uvec3> &&f)
      : meshname(std::move(name)), vertices(std::move(v)), faces(std::move(f)) {}
  const std::vector<Vertex> &getVertices() const override { return vertices; }
  const std::vector<glm::uvec3> &getFaces() const override { return faces; }
  std::string meshname;
  std::vector<Vertex> vertices;
  std::vector<glm::uvec3> faces;
};

// ---- include/loader/ObjLoader.hpp -------------------------------------------------------------------------------
class ObjLoader {
public:
  ObjLoader(const std::string &path, const std::string &meshName, const glm::mat4x4 &model = glm::mat4(1.0f));
  ObjLoader(const std::string &path, const std::string &meshName, const glm::vec3 &axis, float angle,
            const glm::vec3 &translation, const glm::vec3 &scale);
  void setObjFilePath(const std::string &path) { m_path = path; }
  void updateModelMatrix(const glm::vec3 &axis, float angle, const glm::vec3 &translation, const glm::vec3 &scale);
  const glm::mat4x4 &getModelMatrix() { return m_model; }
  std::optional<std::unique_ptr<Mesh>> startLoadingFromFile(const std::string &objName);

private:
  std::string m_path, m_meshName;
  glm::mat4 m_model;
};

class RenderingPipeline;
class TraditionalRasterizer;

// Post-MVP triangle payload (= srz_tri of the C ABI)
struct RasterTriangle {
  float pos[3][3], nrm[3][3], uv[3][2];
};

// ---- include/scene/Scene.hpp ------------------------------------------------------------------------------------
class Scene {
  friend class TraditionalRasterizer;
  friend class RenderingPipeline;

public:
  using ObjTuple = std::tuple<std::shared_ptr<Shader>, std::vector<RasterTriangle>>;

  Scene(const std::string &sceneName, const glm::vec3 &eye, const glm::vec3 &center, const glm::vec3 &up,
        glm::vec3 backgroundColor = glm::vec3(0.f), std::size_t maxdepth = 5, float rr = 0.8f);
  virtual ~Scene() = default;

  const glm::vec3 &loadEyeVec() const { return m_eye; }
  bool setModelMatrix(const std::string &meshName, const glm::vec3 &axis, float angle, const glm::vec3 &translation,
                      const glm::vec3 &scale);
  void setViewMatrix(const glm::vec3 &eye, const glm::vec3 &center, const glm::vec3 &up);
  void setProjectionMatrix(float fovy, float zNear, float zFar);

  bool addGraphicObj(const std::string &path, const std::string &meshName);
  bool addGraphicObj(const std::string &path, const std::string &meshName, const glm::vec3 &axis, float angle,
                     const glm::vec3 &translation, const glm::vec3 &scale);
  bool addGraphicObj(std::unique_ptr<Object> object, const std::string &objectName);
  bool startLoadingMesh(const std::string &meshName);
  std::optional<std::shared_ptr<Object>> getMeshObj(const std::string &meshName);

  bool addShader(const std::string &shaderName, const std::string &texturePath, SHADERS_TYPE type);
  bool addShader(const std::string &shaderName, std::shared_ptr<TextureLoader> text, SHADERS_TYPE type);
  bool bindShader2Mesh(const std::string &meshName, const std::string &shaderName);

  void addLight(std::string name, std::shared_ptr<light_struct> light);
  void addLights(std::vector<std::pair<std::string, std::shared_ptr<light_struct>>> lights);

  // The reference's loadLights() ignores the values given to addLight (src/Scene.cpp:296-312: returns
  // m_lights.size() default-constructed lights).  Default here = the README-documented intent (the registered
  // lights, in registration order); set true to get the as-written behaviour.  DESIGN.md, deviation D9.
  bool reference_exact_lights = false;

  // Vertex stage (src/Scene.cpp:903-964).  Meshes in registration order, triangles in face order (deviation D10).
  std::vector<ObjTuple> loadTriangleStream();
  std::vector<light_struct> loadLights();

  // What loadTriangleStream computes per mesh before touching a vertex (src/Scene.cpp:922-923) — handed to the device
  // vertex stage instead of running the per-face loop on the host.
  struct MeshDraw {
    std::string name;
    Object *mesh;
    std::shared_ptr<Shader> shader;
    glm::mat4 ndc_mvp, normal_m;
  };
  std::vector<MeshDraw> meshDraws();
  float depthScale() const { return scale; }
  float depthOffset() const { return offset; }

  const glm::mat4 &viewMatrix() const { return m_view; }
  const glm::mat4 &projectionMatrix() const { return m_projection; }
  const glm::mat4 &ndcMatrix() const { return m_ndcToScreenMatrix; }

private:
  void setNDCMatrix(std::size_t width, std::size_t height);

  struct ObjInfo {
    std::optional<std::unique_ptr<ObjLoader>> loader;
    std::unique_ptr<Object> mesh;
  };
  std::string m_sceneName;
  std::size_t m_width = 0, m_height = 0;
  float m_aspectRatio = 0.0f;
  glm::vec3 m_eye, m_center, m_up;
  glm::mat4 m_view, m_projection, m_ndcToScreenMatrix;
  float m_fovy = 45.0f, m_near = 0.1f, m_far = 100.0f, scale = 0.0f, offset = 0.0f;
  std::unordered_map<std::string, std::shared_ptr<Shader>> m_shaders;
  std::vector<std::pair<std::string, std::shared_ptr<light_struct>>> m_lights; // registration order
  std::unordered_map<std::string, ObjInfo> m_loadedObjs;
  std::vector<std::string> m_objOrder; // registration order of m_loadedObjs keys
};

// ---- include/base/Render.hpp ------------------------------------------------------------------------------------
enum class Buffers { Color = 1, Depth = 2 };
inline Buffers operator|(Buffers a, Buffers b) { return Buffers((int)a | (int)b); }
inline Buffers operator&(Buffers a, Buffers b) { return Buffers((int)a & (int)b); }
enum class Primitive { LINES, TRIANGLES };

class RenderingPipeline {
public:
  RenderingPipeline();
  RenderingPipeline(std::size_t width, std::size_t height);
  virtual ~RenderingPipeline();

  void clear(Buffers flags);
  // draw + 8-bit resolve (cv::merge + convertTo(CV_8UC3), src/Render.cpp:57-64); no window is opened.
  void display(Primitive type);
  bool addScene(std::shared_ptr<Scene> scene, std::optional<std::string> name = std::nullopt);

  std::size_t width() const { return m_width; }
  std::size_t height() const { return m_height; }
  // The framebuffer lives in HBM (srz_target); these accessors bring the planes to the host when they are stale.
  const std::vector<float> &zBuffer() { syncToHost(); return m_zBuffer; }
  const std::vector<float> &channel(int i) { syncToHost(); return m_channels[i]; } // planar, 0..255 float, plane 0 = texture blue
  const std::vector<uint8_t> &frameBuffer8() const { return m_frameBuffer8; } // interleaved 3 x u8 after display()

  virtual void draw(Primitive type) = 0; // protected in the reference; public here so harnesses can time draw() alone
  void finish();                          // waits for the device to finish what draw() submitted (draw() is asynchronous)

protected:
  void clearFrameBuffer();
  void clearZDepth();
  std::size_t m_width, m_height;
  std::vector<std::pair<std::string, std::shared_ptr<Scene>>> m_scenes; // registration order (deviation D10)
  std::array<std::vector<float>, 3> m_channels;
  std::vector<float> m_zBuffer;
  std::vector<uint8_t> m_frameBuffer8;
  bool m_justCleared = false; // clear(Color|Depth) immediately before draw() → the fused-clear kernel path
  void syncToHost();          // device planes → m_zBuffer / m_channels (no-op when they are current or there is no device)
  srz_ctx *m_ctx = nullptr;       // set by TraditionalRasterizer; a pipeline without it keeps its planes on the host
  srz_target *m_target = nullptr; // device-resident z + colour planes
  bool m_hostStale = false;
};

// ---- include/render/Rasterizer.hpp ------------------------------------------------------------------------------
class TraditionalRasterizer : public RenderingPipeline {
public:
  TraditionalRasterizer();
  TraditionalRasterizer(std::size_t width, std::size_t height);
  ~TraditionalRasterizer() override;
  void draw(Primitive type) override; // throws std::runtime_error("Primitive Type is not supported!") like the reference

  struct Stats {
    uint64_t n_tris = 0, n_culled = 0, pixel_tests = 0, fragments = 0, shaded = 0, visible = 0, visible_textured = 0;
  };
  bool collect_stats = false;
  Stats last_stats;
  // true (default): meshes live on the GPU and draw() runs the vertex stage there (srz_draw_scene);
  // false: Scene::loadTriangleStream() on the host, then srz_draw (same result bit for bit)
  bool device_vertex_stage = true;

private:
  void init();
  std::unordered_map<const Scene *, srz_frameset *> m_sceneSets; // cached 1-frame scenesets (device vertex stage)
  std::unordered_map<const TextureLoader *, int> m_texSlots;
  struct MeshSlot {
    int slot;
    std::size_t n_faces;
    std::uint64_t hash; // of the vertices and faces as uploaded
  };
  std::unordered_map<const Object *, MeshSlot> m_meshSlots; // mesh → its copy on the GPU
  int textureSlot(const std::shared_ptr<Shader> &sh);
  std::vector<std::weak_ptr<TextureLoader>> m_texOwners;
};

} // namespace SoftRasterizer
