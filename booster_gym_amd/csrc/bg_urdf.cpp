// URDF asset loader behind the C ABI (host code, no GPU): what the reference obtains from Isaac Gym's `gym.load_asset(...)` with
// `collapse_fixed_joints` and the asset queries that follow it (envs/t1.py:39-59, 85-108; envs/T1.yaml:61-83), for a host that is not Python.
// booster_gym_amd/utils/urdf.py is the same algorithm on the Python side (and what envs/t1.py of this package uses); tests compare the two.
//
//   parse      <link> (inertial origin/mass/inertia, collision box / cylinder / sphere primitives) and <joint> (type, parent, child, origin,
//              axis, limit) with a small XML reader (elements, attributes, comments, declarations; no entities beyond the five predefined)
//   collapse   links behind `fixed` joints are folded into their parents, leaves first: masses add, centres of mass combine, inertias are
//              rotated into the parent frame and shifted to the combined centre of mass (parallel-axis theorem), collision primitives move along
//   flatten    depth-first body order (URDF child order = Isaac Gym's DoF order, t1.py:57), one revolute joint about +x/+y/+z per non-root
//              body, joint frames not rotated against their parents (true for the T1; anything else is refused with the offending name)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>

#include "bg_model.h"

namespace {

struct V3d { double e[3] = {0, 0, 0}; };
struct M3d { double e[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; };

V3d add(V3d a, V3d b) { V3d o; for (int i = 0; i < 3; i++) o.e[i] = a.e[i] + b.e[i]; return o; }
V3d sub(V3d a, V3d b) { V3d o; for (int i = 0; i < 3; i++) o.e[i] = a.e[i] - b.e[i]; return o; }
V3d scale(double s, V3d a) { V3d o; for (int i = 0; i < 3; i++) o.e[i] = s * a.e[i]; return o; }
double dot(V3d a, V3d b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
V3d mul(const M3d& m, V3d v) { V3d o; for (int i = 0; i < 3; i++) o.e[i] = m.e[i][0] * v.e[0] + m.e[i][1] * v.e[1] + m.e[i][2] * v.e[2]; return o; }
M3d mul(const M3d& a, const M3d& b) {
    M3d o;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o.e[i][j] = a.e[i][0] * b.e[0][j] + a.e[i][1] * b.e[1][j] + a.e[i][2] * b.e[2][j];
    return o;
}
M3d transpose(const M3d& a) { M3d o; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.e[i][j] = a.e[j][i]; return o; }
M3d zero3() { M3d o; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.e[i][j] = 0; return o; }
M3d madd(const M3d& a, const M3d& b) { M3d o; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.e[i][j] = a.e[i][j] + b.e[i][j]; return o; }
bool is_identity(const M3d& a, double tol = 1e-9) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (std::fabs(a.e[i][j] - (i == j ? 1.0 : 0.0)) > tol) return false;
    return true;
}
// URDF rpy = fixed-axis roll (x), pitch (y), yaw (z): R = Rz(y) Ry(p) Rx(r)
M3d rpy_to_mat(V3d rpy) {
    const double cr = std::cos(rpy.e[0]), sr = std::sin(rpy.e[0]), cp = std::cos(rpy.e[1]), sp = std::sin(rpy.e[1]), cy = std::cos(rpy.e[2]), sy = std::sin(rpy.e[2]);
    M3d rx, ry, rz;
    rx.e[1][1] = cr; rx.e[1][2] = -sr; rx.e[2][1] = sr; rx.e[2][2] = cr;
    ry.e[0][0] = cp; ry.e[0][2] = sp; ry.e[2][0] = -sp; ry.e[2][2] = cp;
    rz.e[0][0] = cy; rz.e[0][1] = -sy; rz.e[1][0] = sy; rz.e[1][1] = cy;
    return mul(rz, mul(ry, rx));
}
// inertia about a point displaced by d from the centre of mass: I + m (|d|^2 1 - d d^T)
M3d shifted(const M3d& I, double m, V3d d) {
    M3d o = I;
    const double dd = dot(d, d);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.e[i][j] += m * ((i == j ? dd : 0.0) - d.e[i] * d.e[j]);
    return o;
}

// ------------------------------------------------------------------ minimal XML
struct Xml {
    std::string tag;
    std::map<std::string, std::string> attr;
    std::vector<std::unique_ptr<Xml>> kids;
    const Xml* find(const char* t) const { for (auto& k : kids) if (k->tag == t) return k.get(); return nullptr; }
    std::vector<const Xml*> all(const char* t) const { std::vector<const Xml*> o; for (auto& k : kids) if (k->tag == t) o.push_back(k.get()); return o; }
    const std::string* get(const char* a) const { auto it = attr.find(a); return it == attr.end() ? nullptr : &it->second; }
};
struct XmlReader {
    const std::string& s;
    size_t p = 0;
    explicit XmlReader(const std::string& text) : s(text) {}
    [[noreturn]] void bad(const std::string& why) const { throw std::runtime_error("XML: " + why + " at byte " + std::to_string(p)); }
    bool starts(const char* lit) const { return s.compare(p, std::strlen(lit), lit) == 0; }
    void ws() { while (p < s.size() && std::isspace((unsigned char)s[p])) p++; }
    void skip_to(const char* lit) { size_t q = s.find(lit, p); if (q == std::string::npos) bad(std::string("unterminated ") + lit); p = q + std::strlen(lit); }
    void misc() {  // declarations, comments, doctype, text between elements
        for (;;) {
            while (p < s.size() && s[p] != '<') p++;
            if (p >= s.size()) return;
            if (starts("<?")) skip_to("?>");
            else if (starts("<!--")) skip_to("-->");
            else if (starts("<!")) skip_to(">");
            else return;
        }
    }
    std::string name() {
        size_t q = p;
        while (p < s.size() && (std::isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '-' || s[p] == ':' || s[p] == '.')) p++;
        if (p == q) bad("name expected");
        return s.substr(q, p - q);
    }
    static std::string unescape(const std::string& v) {
        static const char* ent[5][2] = {{"&lt;", "<"}, {"&gt;", ">"}, {"&amp;", "&"}, {"&quot;", "\""}, {"&apos;", "'"}};
        std::string o;
        for (size_t i = 0; i < v.size();) {
            bool hit = false;
            if (v[i] == '&')
                for (auto& e : ent) { size_t n = std::strlen(e[0]); if (v.compare(i, n, e[0]) == 0) { o += e[1]; i += n; hit = true; break; } }
            if (!hit) o += v[i++];
        }
        return o;
    }
    std::unique_ptr<Xml> element(int depth = 0) {
        if (depth > 64) bad("elements nested deeper than 64");  // the recursion below is bounded on malformed / hostile input
        if (p >= s.size() || s[p] != '<') bad("'<' expected");
        p++;
        auto x = std::make_unique<Xml>();
        x->tag = name();
        for (;;) {
            ws();
            if (p >= s.size()) bad("unterminated tag <" + x->tag);
            if (starts("/>")) { p += 2; return x; }
            if (s[p] == '>') { p++; break; }
            std::string a = name();
            ws();
            if (p >= s.size() || s[p] != '=') bad("'=' expected after attribute " + a);
            p++; ws();
            if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) bad("quoted value expected for attribute " + a);
            const char q = s[p++];
            size_t e = s.find(q, p);
            if (e == std::string::npos) bad("unterminated value of attribute " + a);
            x->attr[a] = unescape(s.substr(p, e - p));
            p = e + 1;
        }
        for (;;) {
            misc();
            if (p >= s.size()) bad("missing </" + x->tag + ">");
            if (starts("</")) {
                p += 2;
                if (name() != x->tag) bad("mismatched </...> for <" + x->tag + ">");
                ws();
                if (p >= s.size() || s[p] != '>') bad("'>' expected");
                p++;
                return x;
            }
            x->kids.push_back(element(depth + 1));
        }
    }
    std::unique_ptr<Xml> document() { misc(); auto r = element(); return r; }
};

V3d vec3(const std::string& text, const std::string& what) {
    V3d v;
    if (std::sscanf(text.c_str(), "%lf %lf %lf", &v.e[0], &v.e[1], &v.e[2]) != 3) throw std::runtime_error("expected 3 numbers in " + what + ", got '" + text + "'");
    return v;
}
double num(const Xml* x, const char* a, double dflt) { const std::string* v = x ? x->get(a) : nullptr; return v ? std::atof(v->c_str()) : dflt; }
double num_req(const Xml* x, const char* a, const std::string& what) {
    const std::string* v = x ? x->get(a) : nullptr;
    if (!v) throw std::runtime_error(what + ": attribute '" + a + "' missing");
    return std::atof(v->c_str());
}
void origin(const Xml* parent, V3d* xyz, M3d* rot) {
    *xyz = V3d(); *rot = M3d();
    const Xml* o = parent ? parent->find("origin") : nullptr;
    if (!o) return;
    if (const std::string* v = o->get("xyz")) *xyz = vec3(*v, "origin xyz");
    if (const std::string* v = o->get("rpy")) *rot = rpy_to_mat(vec3(*v, "origin rpy"));
}

struct Shape { int type; double size[3]; V3d pos; M3d rot; };  // type 0 box (size xyz), 1 cylinder (radius, length), 2 sphere (radius)
struct Link { std::string name; double mass = 0; V3d com; M3d inertia = zero3(); std::vector<Shape> shapes; int order = 0; };
struct Joint { std::string name, type, parent, child; V3d xyz, axis; M3d rot; bool has_limit = false; double lower = 0, upper = 0, effort = 0, velocity = 0; };

// rigidly attach `c` (pose xyz / rot in the parent frame) to `p`
void merge(Link& p, const Link& c, V3d xyz, const M3d& rot) {
    const V3d ccom = add(xyz, mul(rot, c.com));
    const M3d cin = mul(rot, mul(c.inertia, transpose(rot)));
    const double m = p.mass + c.mass;
    if (m > 0.0) {
        const V3d com = scale(1.0 / m, add(scale(p.mass, p.com), scale(c.mass, ccom)));
        p.inertia = madd(shifted(p.inertia, p.mass, sub(p.com, com)), shifted(cin, c.mass, sub(ccom, com)));
        p.mass = m; p.com = com;
    }
    for (const Shape& s : c.shapes) {
        Shape t = s;
        t.pos = add(xyz, mul(rot, s.pos));
        t.rot = mul(rot, s.rot);
        p.shapes.push_back(t);
    }
}

}  // namespace

extern "C" int bg_model_load_urdf(const char* path, const bg_asset_options* opt, bg_model** out) {
    if (!path || !opt || !out) return bg_set_error(-1, "bg_model_load_urdf: null argument");
    try {
        std::string text;
        {
            FILE* f = std::fopen(path, "rb");
            if (!f) return bg_set_error(-3, (std::string("bg_model_load_urdf: cannot open ") + path).c_str());
            char buf[65536];
            size_t n;
            while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
            std::fclose(f);
        }
        XmlReader rd(text);
        std::unique_ptr<Xml> robot = rd.document();
        if (robot->tag != "robot") throw std::runtime_error("root element is <" + robot->tag + ">, expected <robot>");

        std::map<std::string, Link> links;
        int order = 0;
        for (const Xml* le : robot->all("link")) {
            Link lk;
            const std::string* nm = le->get("name");
            if (!nm) throw std::runtime_error("<link> without a name");
            lk.name = *nm; lk.order = order++;
            if (const Xml* ine = le->find("inertial")) {
                V3d xyz; M3d rot;
                origin(ine, &xyz, &rot);
                lk.mass = num_req(ine->find("mass"), "value", "link " + lk.name + " <mass>");
                const Xml* it = ine->find("inertia");
                M3d t = zero3();
                t.e[0][0] = num(it, "ixx", 0); t.e[1][1] = num(it, "iyy", 0); t.e[2][2] = num(it, "izz", 0);
                t.e[0][1] = t.e[1][0] = num(it, "ixy", 0); t.e[0][2] = t.e[2][0] = num(it, "ixz", 0); t.e[1][2] = t.e[2][1] = num(it, "iyz", 0);
                lk.com = xyz;
                lk.inertia = mul(rot, mul(t, transpose(rot)));
            }
            for (const Xml* ce : le->all("collision")) {
                Shape s{};
                origin(ce, &s.pos, &s.rot);
                const Xml* g = ce->find("geometry");
                if (!g) continue;
                if (const Xml* b = g->find("box")) {
                    const std::string* sz = b->get("size");
                    if (!sz) throw std::runtime_error("link " + lk.name + ": <box> without size");
                    V3d v = vec3(*sz, "box size");
                    s.type = 0; s.size[0] = v.e[0]; s.size[1] = v.e[1]; s.size[2] = v.e[2];
                } else if (const Xml* c = g->find("cylinder")) {
                    s.type = 1; s.size[0] = num_req(c, "radius", "cylinder"); s.size[1] = num_req(c, "length", "cylinder"); s.size[2] = 0;
                } else if (const Xml* sp = g->find("sphere")) {
                    s.type = 2; s.size[0] = num_req(sp, "radius", "sphere"); s.size[1] = s.size[2] = 0;
                } else {
                    continue;  // mesh collisions carry no primitive data: ignored
                }
                lk.shapes.push_back(s);
            }
            links[lk.name] = lk;
        }
        std::vector<Joint> joints;
        std::map<std::string, bool> is_child;
        for (const Xml* je : robot->all("joint")) {
            Joint j;
            const std::string* nm = je->get("name");
            const std::string* ty = je->get("type");
            const Xml* pe = je->find("parent");
            const Xml* ce = je->find("child");
            if (!nm || !ty || !pe || !ce || !pe->get("link") || !ce->get("link")) throw std::runtime_error("<joint> needs name, type, <parent link>, <child link>");
            j.name = *nm; j.type = *ty; j.parent = *pe->get("link"); j.child = *ce->get("link");
            if (!links.count(j.parent) || !links.count(j.child)) throw std::runtime_error("joint " + j.name + ": unknown link");
            origin(je, &j.xyz, &j.rot);
            j.axis.e[0] = 1.0;
            if (const Xml* ax = je->find("axis")) if (const std::string* v = ax->get("xyz")) j.axis = vec3(*v, "axis xyz");
            if (const Xml* lim = je->find("limit")) {
                j.has_limit = true;
                j.lower = num(lim, "lower", 0); j.upper = num(lim, "upper", 0); j.effort = num(lim, "effort", 0); j.velocity = num(lim, "velocity", 0);
            }
            if (is_child.count(j.child)) throw std::runtime_error("link " + j.child + " is the child of more than one joint (joint " + j.name + ")");
            if (j.child == j.parent) throw std::runtime_error("joint " + j.name + " connects link " + j.child + " to itself");
            joints.push_back(j);
            is_child[j.child] = true;
        }
        std::vector<std::string> roots;
        for (auto& kv : links) if (!is_child.count(kv.first)) roots.push_back(kv.first);
        if (roots.size() != 1) throw std::runtime_error("URDF must have exactly one root link, found " + std::to_string(roots.size()));

        std::map<std::string, std::vector<Joint>> by_parent;
        for (const Joint& j : joints) by_parent[j.parent].push_back(j);
        {   // every link hangs off the root: with one parent joint per link, a link that is not reached sits on a cycle of joints.  (This walk
            // is iterative; the recursive folds below then run on a tree, their depth bounded by the number of links.)
            std::vector<std::string> todo{roots[0]};
            size_t seen = 0;
            while (!todo.empty()) {
                const std::string nm = todo.back(); todo.pop_back(); seen++;
                if (by_parent.count(nm)) for (const Joint& j : by_parent[nm]) todo.push_back(j.child);
            }
            if (seen != links.size()) throw std::runtime_error(std::to_string(links.size() - seen) + " link(s) are not connected to the root link " + roots[0] + " (a cycle of joints)");
        }
        if (opt->collapse_fixed_joints) {
            // fold leaves first so that chains of fixed joints accumulate correctly
            std::function<void(const std::string&)> fold = [&](const std::string& name) {
                std::vector<Joint> mine = by_parent[name];  // copy: the list is edited below
                for (const Joint& j : mine) {
                    fold(j.child);
                    if (j.type != "fixed") continue;
                    merge(links[name], links[j.child], j.xyz, j.rot);
                    auto& lst = by_parent[name];
                    for (size_t k = 0; k < lst.size(); k++) if (lst[k].name == j.name) { lst.erase(lst.begin() + k); break; }
                    std::vector<Joint> grand = by_parent[j.child];  // re-parent the grandchildren through the fixed transform
                    by_parent.erase(j.child);
                    for (Joint g : grand) {
                        g.xyz = add(j.xyz, mul(j.rot, g.xyz));
                        g.rot = mul(j.rot, g.rot);
                        g.parent = name;
                        by_parent[name].push_back(g);
                    }
                }
            };
            fold(roots[0]);
        }

        auto m = std::make_unique<bg_model>();
        bg_model_desc& d = m->desc;
        std::memset(&d, 0, sizeof d);
        std::vector<const Link*> body_link;
        std::function<void(const std::string&, int, const Joint*)> visit = [&](const std::string& name, int par, const Joint* j) {
            const int idx = (int)m->body_names.size();
            if (idx >= BG_NUM_BODIES) throw std::runtime_error("more than " + std::to_string(BG_NUM_BODIES) + " bodies after collapsing fixed joints");
            m->body_names.push_back(name);
            body_link.push_back(&links[name]);
            d.parent[idx] = par;
            if (!j) {
                d.joint_axis[idx] = 0;
            } else {
                if (j->type != "revolute" && j->type != "continuous") throw std::runtime_error("joint " + j->name + ": type " + j->type + " unsupported (revolute / fixed only)");
                if (!is_identity(j->rot)) throw std::runtime_error("joint " + j->name + ": rotated joint frames are unsupported");
                int k = 0;
                for (int a = 1; a < 3; a++) if (std::fabs(j->axis.e[a]) > std::fabs(j->axis.e[k])) k = a;
                for (int a = 0; a < 3; a++)
                    if (std::fabs(j->axis.e[a] - (a == k ? 1.0 : 0.0)) > 1e-9) throw std::runtime_error("joint " + j->name + ": axis must be +x, +y or +z");
                d.joint_axis[idx] = k + 1;
                for (int a = 0; a < 3; a++) d.body_pos[idx][a] = (float)j->xyz.e[a];
                const int q = (int)m->dof_names.size();
                if (q >= BG_NUM_DOFS) throw std::runtime_error("more than " + std::to_string(BG_NUM_DOFS) + " degrees of freedom");
                if (!j->has_limit) throw std::runtime_error("joint " + j->name + ": <limit> missing");
                m->dof_names.push_back(j->name);
                d.dof_lower[q] = (float)j->lower; d.dof_upper[q] = (float)j->upper; d.dof_velocity[q] = (float)j->velocity; d.dof_effort[q] = (float)j->effort;
            }
            std::vector<Joint> kids = by_parent.count(name) ? by_parent[name] : std::vector<Joint>();
            // URDF order of the child links (Isaac Gym orders DoFs depth-first: t1.py:57); insertion sort keeps it stable
            for (size_t a = 1; a < kids.size(); a++)
                for (size_t b = a; b > 0 && links[kids[b].child].order < links[kids[b - 1].child].order; b--) std::swap(kids[b], kids[b - 1]);
            for (const Joint& cj : kids) visit(cj.child, idx, &cj);
        };
        visit(roots[0], -1, nullptr);
        d.num_bodies = (int32_t)m->body_names.size();
        d.num_dofs = (int32_t)m->dof_names.size();
        for (int b = 0; b < d.num_bodies; b++) {
            const Link& l = *body_link[b];
            d.mass[b] = (float)l.mass;
            for (int a = 0; a < 3; a++) d.com[b][a] = (float)l.com.e[a];
            const M3d& t = l.inertia;
            const double i6[6] = {t.e[0][0], t.e[1][1], t.e[2][2], t.e[0][1], t.e[0][2], t.e[1][2]};
            for (int a = 0; a < 6; a++) d.inertia[b][a] = (float)i6[a];
        }
        for (int c = 0; c < 4; c++) for (int a = 0; a < 3; a++) d.feet_edge_pos[c][a] = opt->feet_edge_pos[c][a];
        // contact spheres of the non-foot collision primitives, sorted by body (bodies are visited in index order)
        int foot[2] = {-1, -1};
        for (int f = 0; f < 2; f++)
            if (opt->foot_names[f])
                for (int b = 0; b < d.num_bodies; b++) if (m->body_names[b] == opt->foot_names[f]) foot[f] = b;
        if (opt->body_contacts) {
            for (int b = 0; b < d.num_bodies; b++) {
                if (b == foot[0] || b == foot[1]) continue;
                for (const Shape& s : body_link[b]->shapes) {
                    if (!is_identity(s.rot)) throw std::runtime_error("link " + m->body_names[b] + ": rotated collision primitives are unsupported");
                    auto put = [&](double x, double y, double z, double r) {
                        if (d.num_body_spheres >= BG_MAX_BODY_SPHERES) throw std::runtime_error("more than " + std::to_string(BG_MAX_BODY_SPHERES) + " contact spheres");
                        const int k = d.num_body_spheres++;
                        d.sphere_body[k] = b; d.sphere_pos[k][0] = (float)x; d.sphere_pos[k][1] = (float)y; d.sphere_pos[k][2] = (float)z; d.sphere_radius[k] = (float)r;
                    };
                    if (s.type == 0) {
                        for (int i = -1; i <= 1; i += 2) for (int j = -1; j <= 1; j += 2) for (int k = -1; k <= 1; k += 2)
                            put(s.pos.e[0] + 0.5 * i * s.size[0], s.pos.e[1] + 0.5 * j * s.size[1], s.pos.e[2] + 0.5 * k * s.size[2], 0.0);
                    } else if (s.type == 1) {
                        const double r = s.size[0], half = 0.5 * s.size[1];
                        for (int sg = -1; sg <= 1; sg += 2) put(s.pos.e[0], s.pos.e[1], s.pos.e[2] + sg * std::fmax(half - r, 0.0), r);
                    }
                }
            }
        }
        // self-collision capsules (create_actor(..., self_collisions), envs/t1.py:128): per leg the shank = the link two above the foot (the capsule
        // inscribed in its z-axis cylinder) and the foot (its box, L >= W >= H along x, y, z: radius W / 2, half length L / 2 - H about the box centre)
        // An asset WITHOUT that geometry on either leg is loaded with all capsule radii 0 = "no self-collision geometry": bg_env_create then leaves the
        // leg-against-leg contacts off (include/booster_gym_amd.h).  Geometry that is there but malformed (two cylinders, a rotated primitive, a box
        // that is not longest along x) stays an error.
        bool have_caps = opt->self_collisions && foot[0] >= 0 && foot[1] >= 0;
        for (int leg = 0; leg < 2 && have_caps; leg++) {
            const int fb = foot[leg], ank = d.parent[fb], shank = ank >= 0 ? d.parent[ank] : -1;
            bool cyl = false, box = false;
            if (shank >= 0) for (const Shape& s : body_link[shank]->shapes) cyl = cyl || s.type == 1;
            for (const Shape& s : body_link[fb]->shapes) box = box || s.type == 0;
            have_caps = shank >= 0 && cyl && box;
        }
        if (have_caps) {
            for (int leg = 0; leg < 2; leg++) {
                const int fb = foot[leg], ank = d.parent[fb], shank = ank >= 0 ? d.parent[ank] : -1;
                if (shank < 0) throw std::runtime_error("self-collision capsules: the foot needs a link two above it");
                const Shape *cyl = nullptr, *box = nullptr;
                for (const Shape& s : body_link[shank]->shapes) if (s.type == 1) { if (cyl) throw std::runtime_error("self-collision capsules: more than one cylinder on a shank"); cyl = &s; }
                for (const Shape& s : body_link[fb]->shapes) if (s.type == 0) { if (box) throw std::runtime_error("self-collision capsules: more than one box on a foot"); box = &s; }
                if (!cyl || !box) throw std::runtime_error("self-collision capsules need one cylinder on each shank and one box on each foot");
                if (!is_identity(cyl->rot) || !is_identity(box->rot)) throw std::runtime_error("self-collision capsules: rotated collision primitives are unsupported");
                if (!(box->size[0] >= box->size[1] && box->size[1] >= box->size[2])) throw std::runtime_error("self-collision capsules: the foot box must be longest along x and thinnest along z");
                const double hs = std::fmax(0.5 * cyl->size[1] - cyl->size[0], 0.0), hf = std::fmax(0.5 * box->size[0] - box->size[2], 0.0);
                for (int a = 0; a < 3; a++) {
                    d.self_capsule_a[leg][0][a] = (float)(cyl->pos.e[a] - (a == 2 ? hs : 0.0)); d.self_capsule_b[leg][0][a] = (float)(cyl->pos.e[a] + (a == 2 ? hs : 0.0));
                    d.self_capsule_a[leg][1][a] = (float)(box->pos.e[a] - (a == 0 ? hf : 0.0)); d.self_capsule_b[leg][1][a] = (float)(box->pos.e[a] + (a == 0 ? hf : 0.0));
                }
                d.self_capsule_r[leg][0] = (float)cyl->size[0];
                d.self_capsule_r[leg][1] = (float)(0.5 * box->size[1]);
            }
        }
        const int rc = bg_model_validate(&d);
        if (rc) return rc;
        *out = m.release();
        return 0;
    } catch (const std::exception& ex) {
        return bg_set_error(-1, (std::string("bg_model_load_urdf: ") + path + ": " + ex.what()).c_str());
    }
}

extern "C" const char* bg_model_body_name(const bg_model* m, int32_t i) {
    return (m && i >= 0 && i < (int32_t)m->body_names.size()) ? m->body_names[i].c_str() : nullptr;
}
extern "C" const char* bg_model_dof_name(const bg_model* m, int32_t j) {
    return (m && j >= 0 && j < (int32_t)m->dof_names.size()) ? m->dof_names[j].c_str() : nullptr;
}
extern "C" int32_t bg_model_find_body(const bg_model* m, const char* name) {
    if (!m || !name) return -1;
    for (size_t i = 0; i < m->body_names.size(); i++) if (m->body_names[i] == name) return (int32_t)i;
    return -1;
}
