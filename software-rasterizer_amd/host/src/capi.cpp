// capi.cpp — plain-C access to the C++ host layer (Scene building / vertex stage / asset loaders) for Python tools
// (bench.py, tests).  This is host logic ABOVE the raster boundary: nothing here touches the GPU.  Handles are opaque.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

#include "SoftRasterizer.hpp"

using namespace SoftRasterizer;

namespace SoftRasterizer {
// friend-free access to the private setNDCMatrix: the pipeline does this in addScene (src/Render.cpp:84)
struct HostPipeline : public RenderingPipeline {
  HostPipeline(std::size_t w, std::size_t h) : RenderingPipeline(w, h) {}
  void draw(Primitive) override {}
};
} // namespace SoftRasterizer

struct srzh_scene {
  std::shared_ptr<Scene> scene;
  std::unique_ptr<HostPipeline> pipe;
  std::vector<Scene::ObjTuple> stream;
  std::string err;
};

static glm::vec3 v3(const float *p) { return glm::vec3(p[0], p[1], p[2]); }

extern "C" {

srzh_scene *srzh_scene_create(const char *name, const float *eye, const float *center, const float *up, int width, int height) {
  auto *s = new srzh_scene();
  s->scene = std::make_shared<Scene>(name ? name : "scene", v3(eye), v3(center), v3(up));
  s->pipe = std::make_unique<HostPipeline>((size_t)width, (size_t)height);
  if (!s->pipe->addScene(s->scene)) {
    delete s;
    return nullptr;
  }
  return s;
}
void srzh_scene_destroy(srzh_scene *s) { delete s; }

int srzh_add_obj(srzh_scene *s, const char *path, const char *name, const float *axis, float angle, const float *t, const float *sc) {
  if (!s->scene->addGraphicObj(path, name, v3(axis), angle, v3(t), v3(sc))) return -1;
  return s->scene->startLoadingMesh(name) ? 0 : -1;
}
// texture: file path, or (bgr,w,h) from memory when path == NULL
int srzh_add_shader(srzh_scene *s, const char *name, const char *tex_path, const uint8_t *bgr, int w, int h, int type) {
  try {
    if (tex_path) return s->scene->addShader(name, std::string(tex_path), (SHADERS_TYPE)type) ? 0 : -1;
    return s->scene->addShader(name, std::make_shared<TextureLoader>(bgr, w, h), (SHADERS_TYPE)type) ? 0 : -1;
  } catch (const std::exception &) {
    return -1;
  }
}
int srzh_bind(srzh_scene *s, const char *mesh, const char *shader) { return s->scene->bindShader2Mesh(mesh, shader) ? 0 : -1; }
int srzh_add_light(srzh_scene *s, const char *name, const float *pos, const float *intensity) {
  s->scene->addLight(name, std::make_shared<light_struct>(v3(pos), v3(intensity)));
  return 0;
}
int srzh_set_model(srzh_scene *s, const char *mesh, const float *axis, float angle, const float *t, const float *sc) {
  return s->scene->setModelMatrix(mesh, v3(axis), angle, v3(t), v3(sc)) ? 0 : -1;
}
void srzh_set_view(srzh_scene *s, const float *eye, const float *center, const float *up) { s->scene->setViewMatrix(v3(eye), v3(center), v3(up)); }
void srzh_set_projection(srzh_scene *s, float fovy, float zn, float zf) { s->scene->setProjectionMatrix(fovy, zn, zf); }
void srzh_get_matrices(srzh_scene *s, float *view16, float *proj16, float *ndc16) {
  std::memcpy(view16, s->scene->viewMatrix().data(), 64);
  std::memcpy(proj16, s->scene->projectionMatrix().data(), 64);
  std::memcpy(ndc16, s->scene->ndcMatrix().data(), 64);
}
int srzh_get_model(srzh_scene *s, const char *mesh, float *m16) {
  auto o = s->scene->getMeshObj(mesh);
  if (!o) return -1;
  std::memcpy(m16, (*o)->getModelMatrix().data(), 64);
  return 0;
}

// mesh data as loaded (for checking the OBJ loader): verts (nV,8) = pos3 nrm3 uv2 ; faces (nF,3) uint32
int srzh_mesh_counts(srzh_scene *s, const char *mesh, uint32_t *nv, uint32_t *nf) {
  auto o = s->scene->getMeshObj(mesh);
  if (!o) return -1;
  *nv = (uint32_t)(*o)->getVertices().size(), *nf = (uint32_t)(*o)->getFaces().size();
  return 0;
}
int srzh_mesh_copy(srzh_scene *s, const char *mesh, float *verts8, uint32_t *faces3) {
  auto o = s->scene->getMeshObj(mesh);
  if (!o) return -1;
  const auto &V = (*o)->getVertices();
  const auto &F = (*o)->getFaces();
  for (size_t i = 0; i < V.size(); ++i) {
    float *d = verts8 + i * 8;
    d[0] = V[i].position.x, d[1] = V[i].position.y, d[2] = V[i].position.z;
    d[3] = V[i].normal.x, d[4] = V[i].normal.y, d[5] = V[i].normal.z, d[6] = V[i].texCoord.x, d[7] = V[i].texCoord.y;
  }
  for (size_t i = 0; i < F.size(); ++i) faces3[i * 3] = F[i].x, faces3[i * 3 + 1] = F[i].y, faces3[i * 3 + 2] = F[i].z;
  return 0;
}

// vertex stage: Scene::loadTriangleStream() → batches kept inside the handle until the next call
int srzh_build_stream(srzh_scene *s) {
  s->stream = s->scene->loadTriangleStream();
  return (int)s->stream.size();
}
int srzh_batch_info(srzh_scene *s, int b, int *shader, uint32_t *n_tris, const uint8_t **tex_bgr, int *tex_w, int *tex_h) {
  if (b < 0 || b >= (int)s->stream.size()) return -1;
  const auto &sh = std::get<0>(s->stream[b]);
  *n_tris = (uint32_t)std::get<1>(s->stream[b]).size();
  *shader = sh ? (int)sh->type() : -1;
  if (sh && sh->getTextureObject()) {
    *tex_bgr = sh->getTextureObject()->bgr().data();
    *tex_w = (int)sh->getTextureObject()->width(), *tex_h = (int)sh->getTextureObject()->height();
  } else {
    *tex_bgr = nullptr, *tex_w = *tex_h = 0;
  }
  return 0;
}
int srzh_batch_copy(srzh_scene *s, int b, void *out_tris96) {
  if (b < 0 || b >= (int)s->stream.size()) return -1;
  const auto &t = std::get<1>(s->stream[b]);
  std::memcpy(out_tris96, t.data(), t.size() * sizeof(RasterTriangle));
  return 0;
}
// per-mesh matrices of the vertex stage (what the device vertex stage consumes), in registration order
int srzh_n_mesh_draws(srzh_scene *s) { return (int)s->scene->meshDraws().size(); }
int srzh_mesh_draw(srzh_scene *s, int i, char *name64, int *shader, float *ndc_mvp16, float *normal16, float *zscale_offset2) {
  auto d = s->scene->meshDraws();
  if (i < 0 || i >= (int)d.size()) return -1;
  std::snprintf(name64, 64, "%s", d[i].name.c_str());
  *shader = d[i].shader ? (int)d[i].shader->type() : -1;
  std::memcpy(ndc_mvp16, d[i].ndc_mvp.data(), 64);
  std::memcpy(normal16, d[i].normal_m.data(), 64);
  zscale_offset2[0] = s->scene->depthScale(), zscale_offset2[1] = s->scene->depthOffset();
  return 0;
}
int srzh_n_lights(srzh_scene *s) { return (int)s->scene->loadLights().size(); }
int srzh_lights_copy(srzh_scene *s, float *out6) {
  auto L = s->scene->loadLights();
  for (size_t i = 0; i < L.size(); ++i) {
    out6[i * 6] = L[i].position.x, out6[i * 6 + 1] = L[i].position.y, out6[i * 6 + 2] = L[i].position.z;
    out6[i * 6 + 3] = L[i].intensity.x, out6[i * 6 + 4] = L[i].intensity.y, out6[i * 6 + 5] = L[i].intensity.z;
  }
  return (int)L.size();
}
void srzh_set_reference_exact_lights(srzh_scene *s, int on) { s->scene->reference_exact_lights = on != 0; }
void srzh_get_shader_constants(float *ka3, float *ks3, float *p_kh_kn) {
  ka3[0] = Shader::ka.x, ka3[1] = Shader::ka.y, ka3[2] = Shader::ka.z;
  ks3[0] = Shader::ks.x, ks3[1] = Shader::ks.y, ks3[2] = Shader::ks.z;
  p_kh_kn[0] = Shader::p, p_kh_kn[1] = Shader::kh, p_kh_kn[2] = Shader::kn;
}

// image decode (cv::imread stand-in) for checking against PIL: returns 0 and w,h; then copy
int srzh_image_size(const char *path, int *w, int *h) {
  try {
    TextureLoader t{std::string(path)};
    *w = (int)t.width(), *h = (int)t.height();
    return 0;
  } catch (const std::exception &) {
    return -1;
  }
}
int srzh_image_copy(const char *path, uint8_t *out_bgr) {
  try {
    TextureLoader t{std::string(path)};
    std::memcpy(out_bgr, t.bgr().data(), t.bgr().size());
    return 0;
  } catch (const std::exception &) {
    return -1;
  }
}

} // extern "C"
