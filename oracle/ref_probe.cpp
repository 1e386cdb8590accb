// ref_probe.cpp — compiles the reference's VENDORED third-party headers where they lie under
// /root/reference (lib/tiny_obj_loader.h v2.0.0, rdn/glm 0.9.8.5) and dumps what they produce, so
// that tests can pin our own OBJ/MTL reader and lookAt against them.  Test infrastructure only;
// built into oracle/_ref/ by oracle/Makefile, never linked into the product.
//
// usage: ref_probe obj <file.obj> <mtl_dir>   -> JSON on stdout (raw tinyobj parse)
//        ref_probe lookat ex ey ez cx cy cz ux uy uz -> 16 floats (glm column-major)
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"
#include "glm/glm.hpp"
#include "glm/gtc/matrix_transform.hpp"
#include "glm/gtc/type_ptr.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>

static void arr(const char* k, const float* v, int n, bool comma = true) {
    printf("\"%s\":[", k);
    for (int i = 0; i < n; i++) printf("%s%.9g", i ? "," : "", v[i]);
    printf("]%s", comma ? "," : "");
}
int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "obj")) {
        tinyobj::ObjReaderConfig cfg; cfg.mtl_search_path = argv[3];   // same call shape as ObjLoader.h:394-405
        tinyobj::ObjReader reader;
        if (!reader.ParseFromFile(argv[2], cfg)) { fprintf(stderr, "parse error: %s\n", reader.Error().c_str()); return 1; }
        const auto& at = reader.GetAttrib(); const auto& sh = reader.GetShapes(); const auto& mt = reader.GetMaterials();
        printf("{");
        arr("vertices", at.vertices.data(), (int)at.vertices.size());
        arr("normals", at.normals.data(), (int)at.normals.size());
        printf("\"materials\":[");
        for (size_t i = 0; i < mt.size(); i++) {
            const auto& m = mt[i];
            printf("%s{\"name\":\"%s\",", i ? "," : "", m.name.c_str());
            arr("diffuse", m.diffuse, 3); arr("specular", m.specular, 3); arr("emission", m.emission, 3);
            arr("ambient", m.ambient, 3); arr("transmittance", m.transmittance, 3);
            // texture names in the order of royaltracer-dx_amd/host/ObjLoader.h MapSlot (Ka Kd Ks Ke Ns bump d disp refl Pr Pm Ps norm)
            const std::string* tex[13] = {&m.ambient_texname, &m.diffuse_texname, &m.specular_texname, &m.emissive_texname, &m.specular_highlight_texname, &m.bump_texname,
                                          &m.alpha_texname, &m.displacement_texname, &m.reflection_texname, &m.roughness_texname, &m.metallic_texname, &m.sheen_texname, &m.normal_texname};
            printf("\"tex\":[");
            for (int k = 0; k < 13; k++) printf("%s\"%s\"", k ? "," : "", tex[k]->c_str());
            printf("],\"shininess\":%.9g,\"illum\":%d,\"clearcoat_roughness\":%.9g,\"anisotropy\":%.9g,\"anisotropy_rotation\":%.9g,", m.shininess, m.illum, m.clearcoat_roughness,
                   m.anisotropy, m.anisotropy_rotation);
            printf("\"dissolve\":%.9g,\"roughness\":%.9g,\"metallic\":%.9g,\"sheen\":%.9g,\"clearcoat_thickness\":%.9g,\"ior\":%.9g}",
                   m.dissolve, m.roughness, m.metallic, m.sheen, m.clearcoat_thickness, m.ior);
        }
        printf("],\"shapes\":[");
        for (size_t s = 0; s < sh.size(); s++) {
            const auto& me = sh[s].mesh;
            printf("%s{\"name\":\"%s\",\"num_face_vertices\":[", s ? "," : "", sh[s].name.c_str());
            for (size_t i = 0; i < me.num_face_vertices.size(); i++) printf("%s%d", i ? "," : "", (int)me.num_face_vertices[i]);
            printf("],\"material_ids\":[");
            for (size_t i = 0; i < me.material_ids.size(); i++) printf("%s%d", i ? "," : "", me.material_ids[i]);
            printf("],\"vertex_index\":[");
            for (size_t i = 0; i < me.indices.size(); i++) printf("%s%d", i ? "," : "", me.indices[i].vertex_index);
            printf("],\"normal_index\":[");
            for (size_t i = 0; i < me.indices.size(); i++) printf("%s%d", i ? "," : "", me.indices[i].normal_index);
            printf("]}");
        }
        printf("]}\n");
        return 0;
    }
    if (argc >= 11 && !strcmp(argv[1], "lookat")) {
        float a[9]; for (int i = 0; i < 9; i++) a[i] = (float)atof(argv[2 + i]);
        glm::mat4 m = glm::lookAt(glm::vec3(a[0], a[1], a[2]), glm::vec3(a[3], a[4], a[5]), glm::vec3(a[6], a[7], a[8]));  // manipulator.cpp:307
        const float* p = glm::value_ptr(m);
        for (int i = 0; i < 16; i++) printf("%s%.9g", i ? " " : "", p[i]);
        printf("\n");
        return 0;
    }
    fprintf(stderr, "usage: ref_probe obj <file.obj> <mtl_dir> | lookat ex ey ez cx cy cz ux uy uz\n");
    return 2;
}
