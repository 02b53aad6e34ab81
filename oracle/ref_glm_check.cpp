// ref_glm_check.cpp -- TEST INFRASTRUCTURE. Thin extern "C" harness around the reference's own vendored,
// header-only glm (RAST/third_party/glm, version 0.9.9.9), included from where it lies under
// /root/reference at build time (never copied into this repo). tests/test_oracle_pins.py compares the
// plain-C matrix helpers of gof_oracle.c against these, bit for bit, which pins the evaluation orders
// (mat3*mat3, mat4*mat4, mat3*vec3, vec3*mat3, length, transpose) that forward.cu relies on.
#include <glm/glm.hpp>
#include <cstring>

extern "C" {
void ref_glm_m3_mul(const float* a, const float* b, float* out)
{
    glm::mat3 A, B; std::memcpy(&A, a, sizeof A); std::memcpy(&B, b, sizeof B);
    glm::mat3 R = A * B; std::memcpy(out, &R, sizeof R);
}
void ref_glm_m4_mul(const float* a, const float* b, float* out)
{
    glm::mat4 A, B; std::memcpy(&A, a, sizeof A); std::memcpy(&B, b, sizeof B);
    glm::mat4 R = A * B; std::memcpy(out, &R, sizeof R);
}
void ref_glm_m3_mul_v(const float* a, const float* v, float* out)
{
    glm::mat3 A; glm::vec3 V; std::memcpy(&A, a, sizeof A); std::memcpy(&V, v, sizeof V);
    glm::vec3 R = A * V; std::memcpy(out, &R, sizeof R);
}
void ref_glm_v_mul_m3(const float* v, const float* a, float* out)
{
    glm::mat3 A; glm::vec3 V; std::memcpy(&A, a, sizeof A); std::memcpy(&V, v, sizeof V);
    glm::vec3 R = V * A; std::memcpy(out, &R, sizeof R);
}
// transpose(T) * transpose(Vrk) * T, the exact expression shape of forward.cu:107
void ref_glm_tvt(const float* t, const float* vrk, float* out)
{
    glm::mat3 T, V; std::memcpy(&T, t, sizeof T); std::memcpy(&V, vrk, sizeof V);
    glm::mat3 R = glm::transpose(T) * glm::transpose(V) * T; std::memcpy(out, &R, sizeof R);
}
// -M * v (unary minus binds to the matrix), the expression shape of forward.cu:226
void ref_glm_neg_m3_mul_v(const float* a, const float* v, float* out)
{
    glm::mat3 A; glm::vec3 V; std::memcpy(&A, a, sizeof A); std::memcpy(&V, v, sizeof V);
    glm::vec3 R = -A * V; std::memcpy(out, &R, sizeof R);
}
// dir / glm::length(dir), forward.cu:27
void ref_glm_normalize_by_length(const float* v, float* out)
{
    glm::vec3 V; std::memcpy(&V, v, sizeof V);
    glm::vec3 R = V / glm::length(V); std::memcpy(out, &R, sizeof R);
}
}
