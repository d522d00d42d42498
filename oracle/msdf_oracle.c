/* TEST INFRASTRUCTURE ONLY -- see msdf_oracle.h. Plain C restatement of the msdfgen v1.13.0 hot path.
 * Every function cites the reference file:line it follows. Arithmetic is written in the reference's operation
 * order; build with -ffp-contract=off (oracle/Makefile). */
#define _GNU_SOURCE
#include "msdf_oracle.h"

#include <stddef.h>
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------ Vector2.hpp */

typedef struct { double x, y; } v2;

static inline v2 V(double x, double y) { v2 r = { x, y }; return r; }
static inline v2 vadd(v2 a, v2 b) { return V(a.x+b.x, a.y+b.y); }              /* Vector2.hpp:131 */
static inline v2 vsub(v2 a, v2 b) { return V(a.x-b.x, a.y-b.y); }              /* Vector2.hpp:135 */
static inline v2 vneg(v2 a) { return V(-a.x, -a.y); }                          /* Vector2.hpp:123 */
static inline v2 vmulv(v2 a, v2 b) { return V(a.x*b.x, a.y*b.y); }             /* Vector2.hpp:139 */
static inline v2 vdivv(v2 a, v2 b) { return V(a.x/b.x, a.y/b.y); }             /* Vector2.hpp:143 */
static inline v2 smul(double a, v2 b) { return V(a*b.x, a*b.y); }              /* Vector2.hpp:147 */
static inline double dot(v2 a, v2 b) { return a.x*b.x+a.y*b.y; }               /* Vector2.hpp:106 */
static inline double cross(v2 a, v2 b) { return a.x*b.y-a.y*b.x; }             /* Vector2.hpp:111 */
static inline double vlen(v2 a) { return sqrt(a.x*a.x+a.y*a.y); }              /* Vector2.hpp:37 */
static inline int vnonzero(v2 a) { return a.x || a.y; }                        /* Vector2.hpp:63 */

static inline v2 vnormalize(v2 a, int allowZero) {                              /* Vector2.hpp:42-46 */
    double len = vlen(a);
    if (len)
        return V(a.x/len, a.y/len);
    return V(0, !allowZero);
}

static inline v2 vorthonormal(v2 a, int polarity, int allowZero) {              /* Vector2.hpp:54-58 */
    double len = vlen(a);
    if (len)
        return polarity ? V(-a.y/len, a.x/len) : V(a.y/len, -a.x/len);
    return polarity ? V(0, !allowZero) : V(0, -!allowZero);
}

/* arithmetics.hpp:27-31 (T = Vector2, S = double) */
static inline v2 vmix(v2 a, v2 b, double w) { return vadd(smul(1.-w, a), smul(w, b)); }
/* arithmetics.hpp:59-61: 1 for positive, -1 for zero and negative */
static inline int nonZeroSign(double n) { return 2*(n > 0)-1; }
static inline int isign(double n) { return (0 < n)-(n < 0); }                  /* arithmetics.hpp:53-55 */
static inline double dmin(double a, double b) { return b < a ? b : a; }         /* arithmetics.hpp:9-11 */
static inline double dmax(double a, double b) { return a < b ? b : a; }         /* arithmetics.hpp:15-17 */
static inline double dmedian(double a, double b, double c) { return dmax(dmin(a, b), dmin(dmax(a, b), c)); } /* :21-23 */
static inline float fmin_(float a, float b) { return b < a ? b : a; }
static inline float fmax_(float a, float b) { return a < b ? b : a; }
static inline float fmedian(float a, float b, float c) { return fmax_(fmin_(a, b), fmin_(fmax_(a, b), c)); }
/* arithmetics.hpp:27-31 (T = float, S = double): computed in double, truncated to float */
static inline float fmix(float a, float b, double w) { return (float) ((1.-w)*a+w*b); }

/* ------------------------------------------------------------------------------------- equation-solver.cpp */

int orc_solve_quadratic(double *x, double a, double b, double c) {             /* equation-solver.cpp:9-32 */
    if (a == 0 || fabs(b) > 1e12*fabs(a)) {
        if (b == 0) {
            if (c == 0)
                return -1;
            return 0;
        }
        x[0] = -c/b;
        return 1;
    }
    double dscr = b*b-4*a*c;
    if (dscr > 0) {
        dscr = sqrt(dscr);
        x[0] = (-b+dscr)/(2*a);
        x[1] = (-b-dscr)/(2*a);
        return 2;
    } else if (dscr == 0) {
        x[0] = -b/(2*a);
        return 1;
    } else
        return 0;
}

static int solveCubicNormed(double *x, double a, double b, double c) {          /* equation-solver.cpp:34-61 */
    double a2 = a*a;
    double q = 1/9.*(a2-3*b);
    double r = 1/54.*(a*(2*a2-9*b)+27*c);
    double r2 = r*r;
    double q3 = q*q*q;
    a *= 1/3.;
    if (r2 < q3) {
        double t = r/sqrt(q3);
        if (t < -1) t = -1;
        if (t > 1) t = 1;
        t = acos(t);
        q = -2*sqrt(q);
        x[0] = q*cos(1/3.*t)-a;
        x[1] = q*cos(1/3.*(t+2*M_PI))-a;
        x[2] = q*cos(1/3.*(t-2*M_PI))-a;
        return 3;
    } else {
        double u = (r < 0 ? 1 : -1)*pow(fabs(r)+sqrt(r2-q3), 1/3.);
        double v = u == 0 ? 0 : q/u;
        x[0] = (u+v)-a;
        if (u == v || fabs(u-v) < 1e-12*fabs(u+v)) {
            x[1] = -.5*(u+v)-a;
            return 2;
        }
        return 1;
    }
}

int orc_solve_cubic(double *x, double a, double b, double c, double d) {       /* equation-solver.cpp:63-70 */
    if (a != 0) {
        double bn = b/a;
        if (fabs(bn) < 1e6)
            return solveCubicNormed(x, bn, c/a, d/a);
    }
    return orc_solve_quadratic(x, b, c, d);
}

/* ------------------------------------------------------------------------------------------ edge-segments.cpp */

typedef struct { double distance, dot; } sdist;                                 /* SignedDistance.hpp:10-20 */

static inline int sd_less(sdist a, sdist b) {                                   /* SignedDistance.hpp:22-24 */
    return fabs(a.distance) < fabs(b.distance) || (fabs(a.distance) == fabs(b.distance) && a.dot < b.dot);
}

typedef struct {
    int type;    /* 1, 2, 3 */
    int color;
    v2 p[4];
} edge_t;

static edge_t load_edge(const orc_shape *s, int e) {
    edge_t r;
    const double *p = s->points+8*(size_t) e;
    r.type = s->types[e];
    r.color = s->colors[e];
    for (int i = 0; i < 4; ++i)
        r.p[i] = V(p[2*i], p[2*i+1]);
    return r;
}

static v2 edge_point(const edge_t *e, double t) {                               /* edge-segments.cpp:108-119 */
    switch (e->type) {
        case 1:
            return vmix(e->p[0], e->p[1], t);
        case 2:
            return vmix(vmix(e->p[0], e->p[1], t), vmix(e->p[1], e->p[2], t), t);
        default: {
            v2 p12 = vmix(e->p[1], e->p[2], t);
            return vmix(vmix(vmix(e->p[0], e->p[1], t), p12, t), vmix(p12, vmix(e->p[2], e->p[3], t), t), t);
        }
    }
}

static v2 edge_direction(const edge_t *e, double t) {                           /* edge-segments.cpp:121-139 */
    switch (e->type) {
        case 1:
            return vsub(e->p[1], e->p[0]);
        case 2: {
            v2 tangent = vmix(vsub(e->p[1], e->p[0]), vsub(e->p[2], e->p[1]), t);
            if (!vnonzero(tangent))
                return vsub(e->p[2], e->p[0]);
            return tangent;
        }
        default: {
            v2 tangent = vmix(vmix(vsub(e->p[1], e->p[0]), vsub(e->p[2], e->p[1]), t), vmix(vsub(e->p[2], e->p[1]), vsub(e->p[3], e->p[2]), t), t);
            if (!vnonzero(tangent)) {
                if (t == 0) return vsub(e->p[2], e->p[0]);
                if (t == 1) return vsub(e->p[3], e->p[1]);
            }
            return tangent;
        }
    }
}

static sdist sd_linear(const v2 *p, v2 origin, double *param) {                 /* edge-segments.cpp:173-185 */
    v2 aq = vsub(origin, p[0]);
    v2 ab = vsub(p[1], p[0]);
    *param = dot(aq, ab)/dot(ab, ab);
    v2 eq = vsub(p[*param > .5], origin);
    double endpointDistance = vlen(eq);
    if (*param > 0 && *param < 1) {
        double orthoDistance = dot(vorthonormal(ab, 0, 0), aq);
        if (fabs(orthoDistance) < endpointDistance) {
            sdist r = { orthoDistance, 0 };
            return r;
        }
    }
    sdist r = { nonZeroSign(cross(aq, ab))*endpointDistance, fabs(dot(vnormalize(ab, 0), vnormalize(eq, 0))) };
    return r;
}

static sdist sd_quadratic(const edge_t *e, v2 origin, double *param) {          /* edge-segments.cpp:187-226 */
    const v2 *p = e->p;
    v2 qa = vsub(p[0], origin);
    v2 ab = vsub(p[1], p[0]);
    v2 br = vsub(vsub(p[2], p[1]), ab);
    double a = dot(br, br);
    double b = 3*dot(ab, br);
    double c = 2*dot(ab, ab)+dot(qa, br);
    double d = dot(qa, ab);
    double t[3];
    int solutions = orc_solve_cubic(t, a, b, c, d);

    v2 epDir = edge_direction(e, 0);
    double minDistance = nonZeroSign(cross(epDir, qa))*vlen(qa);
    *param = -dot(qa, epDir)/dot(epDir, epDir);
    {
        double distance = vlen(vsub(p[2], origin));
        if (distance < fabs(minDistance)) {
            epDir = edge_direction(e, 1);
            minDistance = nonZeroSign(cross(epDir, vsub(p[2], origin)))*distance;
            *param = dot(vsub(origin, p[1]), epDir)/dot(epDir, epDir);
        }
    }
    for (int i = 0; i < solutions; ++i) {
        if (t[i] > 0 && t[i] < 1) {
            v2 qe = vadd(vadd(qa, smul(2*t[i], ab)), smul(t[i]*t[i], br));
            double distance = vlen(qe);
            if (distance <= fabs(minDistance)) {
                minDistance = nonZeroSign(cross(vadd(ab, smul(t[i], br)), qe))*distance;
                *param = t[i];
            }
        }
    }

    sdist r;
    r.distance = minDistance;
    if (*param >= 0 && *param <= 1)
        r.dot = 0;
    else if (*param < .5)
        r.dot = fabs(dot(vnormalize(edge_direction(e, 0), 0), vnormalize(qa, 0)));
    else
        r.dot = fabs(dot(vnormalize(edge_direction(e, 1), 0), vnormalize(vsub(p[2], origin), 0)));
    return r;
}

#define CUBIC_SEARCH_STARTS 4 /* edge-segments.h:11 */
#define CUBIC_SEARCH_STEPS 4  /* edge-segments.h:12 */

static sdist sd_cubic(const edge_t *e, v2 origin, double *param) {              /* edge-segments.cpp:228-277 */
    const v2 *p = e->p;
    v2 qa = vsub(p[0], origin);
    v2 ab = vsub(p[1], p[0]);
    v2 br = vsub(vsub(p[2], p[1]), ab);
    v2 as = vsub(vsub(vsub(p[3], p[2]), vsub(p[2], p[1])), br);

    v2 epDir = edge_direction(e, 0);
    double minDistance = nonZeroSign(cross(epDir, qa))*vlen(qa);
    *param = -dot(qa, epDir)/dot(epDir, epDir);
    {
        double distance = vlen(vsub(p[3], origin));
        if (distance < fabs(minDistance)) {
            epDir = edge_direction(e, 1);
            minDistance = nonZeroSign(cross(epDir, vsub(p[3], origin)))*distance;
            *param = dot(vsub(epDir, vsub(p[3], origin)), epDir)/dot(epDir, epDir);
        }
    }
    for (int i = 0; i <= CUBIC_SEARCH_STARTS; ++i) {
        double t = 1./CUBIC_SEARCH_STARTS*i;
        v2 qe = vadd(vadd(vadd(qa, smul(3*t, ab)), smul(3*t*t, br)), smul(t*t*t, as));
        v2 d1 = vadd(vadd(smul(3, ab), smul(6*t, br)), smul(3*t*t, as));
        v2 d2 = vadd(smul(6, br), smul(6*t, as));
        double improvedT = t-dot(qe, d1)/(dot(d1, d1)+dot(qe, d2));
        if (improvedT > 0 && improvedT < 1) {
            int remainingSteps = CUBIC_SEARCH_STEPS;
            do {
                t = improvedT;
                qe = vadd(vadd(vadd(qa, smul(3*t, ab)), smul(3*t*t, br)), smul(t*t*t, as));
                d1 = vadd(vadd(smul(3, ab), smul(6*t, br)), smul(3*t*t, as));
                if (!--remainingSteps)
                    break;
                d2 = vadd(smul(6, br), smul(6*t, as));
                improvedT = t-dot(qe, d1)/(dot(d1, d1)+dot(qe, d2));
            } while (improvedT > 0 && improvedT < 1);
            double distance = vlen(qe);
            if (distance < fabs(minDistance)) {
                minDistance = nonZeroSign(cross(d1, qe))*distance;
                *param = t;
            }
        }
    }

    sdist r;
    r.distance = minDistance;
    if (*param >= 0 && *param <= 1)
        r.dot = 0;
    else if (*param < .5)
        r.dot = fabs(dot(vnormalize(edge_direction(e, 0), 0), vnormalize(qa, 0)));
    else
        r.dot = fabs(dot(vnormalize(edge_direction(e, 1), 0), vnormalize(vsub(p[3], origin), 0)));
    return r;
}

static sdist edge_signed_distance(const edge_t *e, v2 origin, double *param) {
    switch (e->type) {
        case 1: return sd_linear(e->p, origin, param);
        case 2: return sd_quadratic(e, origin, param);
        default: return sd_cubic(e, origin, param);
    }
}

void orc_signed_distance(int type, const double *p, double ox, double oy, double *out) {
    edge_t e;
    e.type = type;
    e.color = 7;
    for (int i = 0; i < 4; ++i)
        e.p[i] = V(p[2*i], p[2*i+1]);
    double param = 0;
    sdist sd = edge_signed_distance(&e, V(ox, oy), &param);
    out[0] = sd.distance, out[1] = sd.dot, out[2] = param;
}

/* EdgeSegment::distanceToPerpendicularDistance, edge-segments.cpp:28-52 */
static void distance_to_perpendicular(const edge_t *e, sdist *distance, v2 origin, double param) {
    if (param < 0) {
        v2 dir = vnormalize(edge_direction(e, 0), 0);
        v2 aq = vsub(origin, edge_point(e, 0));
        double ts = dot(aq, dir);
        if (ts < 0) {
            double perpendicularDistance = cross(aq, dir);
            if (fabs(perpendicularDistance) <= fabs(distance->distance)) {
                distance->distance = perpendicularDistance;
                distance->dot = 0;
            }
        }
    } else if (param > 1) {
        v2 dir = vnormalize(edge_direction(e, 1), 0);
        v2 bq = vsub(origin, edge_point(e, 1));
        double ts = dot(bq, dir);
        if (ts > 0) {
            double perpendicularDistance = cross(bq, dir);
            if (fabs(perpendicularDistance) <= fabs(distance->distance)) {
                distance->distance = perpendicularDistance;
                distance->dot = 0;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------ Contour.cpp */

static double shoelace(v2 a, v2 b) { return (b.x-a.x)*(a.y+b.y); }               /* Contour.cpp:7-9 */

static int contour_winding(const orc_shape *s, int c) {                          /* Contour.cpp:57-81 */
    int begin = s->contour_offsets[c], end = s->contour_offsets[c+1];
    int n = end-begin;
    if (n == 0)
        return 0;
    double total = 0;
    if (n == 1) {
        edge_t e = load_edge(s, begin);
        v2 a = edge_point(&e, 0), b = edge_point(&e, 1/3.), cc = edge_point(&e, 2/3.);
        total += shoelace(a, b);
        total += shoelace(b, cc);
        total += shoelace(cc, a);
    } else if (n == 2) {
        edge_t e0 = load_edge(s, begin), e1 = load_edge(s, begin+1);
        v2 a = edge_point(&e0, 0), b = edge_point(&e0, .5), cc = edge_point(&e1, 0), d = edge_point(&e1, .5);
        total += shoelace(a, b);
        total += shoelace(b, cc);
        total += shoelace(cc, d);
        total += shoelace(d, a);
    } else {
        edge_t last = load_edge(s, end-1);
        v2 prev = edge_point(&last, 0);
        for (int i = begin; i < end; ++i) {
            edge_t e = load_edge(s, i);
            v2 cur = edge_point(&e, 0);
            total += shoelace(prev, cur);
            prev = cur;
        }
    }
    return isign(total);
}

void orc_contour_windings(const orc_shape *shape, int32_t *windings) {
    for (int c = 0; c < shape->n_contours; ++c)
        windings[c] = contour_winding(shape, c);
}

/* ------------------------------------------------------------------------------------------ edge-selectors.cpp */

#define DISTANCE_DELTA_FACTOR 1.001 /* edge-selectors.cpp:8 */

typedef struct {                    /* PerpendicularDistanceSelectorBase, edge-selectors.h:40-70 */
    sdist minTrue;
    double minNeg, minPos;
    int nearEdge;                   /* flat edge index, -1 = NULL */
    double nearParam;
} perp_base;

typedef struct {
    int kind;                       /* 1 true, 2 perp, 3 multi, 4 multi+true */
    v2 p;
    sdist minDistance;              /* TrueDistanceSelector */
    perp_base ch[3];                /* [0] for kind 2; r, g, b for kinds 3, 4 */
} selector;

static void perp_base_init(perp_base *b) {                                      /* edge-selectors.cpp:54 + SignedDistance.hpp:17 */
    b->minTrue.distance = -DBL_MAX;
    b->minTrue.dot = 0;
    b->minNeg = -fabs(b->minTrue.distance);
    b->minPos = fabs(b->minTrue.distance);
    b->nearEdge = -1;
    b->nearParam = 0;
}

static void perp_base_reset(perp_base *b, double delta) {                       /* edge-selectors.cpp:56-62 */
    b->minTrue.distance += nonZeroSign(b->minTrue.distance)*delta;
    b->minNeg = -fabs(b->minTrue.distance);
    b->minPos = fabs(b->minTrue.distance);
    b->nearEdge = -1;
    b->nearParam = 0;
}

/* Default construction followed by reset(p), as oneShotDistance does (ShapeDistanceFinder.hpp:38-39). */
static void selector_init(selector *s, int kind, v2 p) {
    s->kind = kind;
    s->p = V(0, 0);
    s->minDistance.distance = -DBL_MAX, s->minDistance.dot = 0;
    for (int i = 0; i < 3; ++i)
        perp_base_init(&s->ch[i]);
    double delta = DISTANCE_DELTA_FACTOR*vlen(vsub(p, s->p));                   /* edge-selectors.cpp:13, 124, 167 */
    if (kind == 1)
        s->minDistance.distance += nonZeroSign(s->minDistance.distance)*delta; /* edge-selectors.cpp:15 */
    else
        for (int i = 0; i < (kind == 2 ? 1 : 3); ++i)
            perp_base_reset(&s->ch[i], delta);
    s->p = p;
}

static int get_perpendicular_distance(double *distance, v2 ep, v2 edgeDir) {    /* edge-selectors.cpp:42-52 */
    double ts = dot(ep, edgeDir);
    if (ts > 0) {
        double perpendicularDistance = cross(ep, edgeDir);
        if (fabs(perpendicularDistance) < fabs(*distance)) {
            *distance = perpendicularDistance;
            return 1;
        }
    }
    return 0;
}

static void add_true(perp_base *b, int edge, sdist distance, double param) {    /* edge-selectors.cpp:81-87 */
    if (sd_less(distance, b->minTrue)) {
        b->minTrue = distance;
        b->nearEdge = edge;
        b->nearParam = param;
    }
}

static void add_perp(perp_base *b, double distance) {                           /* edge-selectors.cpp:89-94 */
    if (distance <= 0 && distance > b->minNeg)
        b->minNeg = distance;
    if (distance >= 0 && distance < b->minPos)
        b->minPos = distance;
}

static void perp_base_merge(perp_base *b, const perp_base *o) {                 /* edge-selectors.cpp:96-106 */
    if (sd_less(o->minTrue, b->minTrue)) {
        b->minTrue = o->minTrue;
        b->nearEdge = o->nearEdge;
        b->nearParam = o->nearParam;
    }
    if (o->minNeg > b->minNeg)
        b->minNeg = o->minNeg;
    if (o->minPos < b->minPos)
        b->minPos = o->minPos;
}

static double perp_base_compute(const orc_shape *shape, const perp_base *b, v2 p) { /* edge-selectors.cpp:108-117 */
    double minDistance = b->minTrue.distance < 0 ? b->minNeg : b->minPos;
    if (b->nearEdge >= 0) {
        sdist distance = b->minTrue;
        edge_t e = load_edge(shape, b->nearEdge);
        distance_to_perpendicular(&e, &distance, p, b->nearParam);
        if (fabs(distance.distance) < fabs(minDistance))
            minDistance = distance.distance;
    }
    return minDistance;
}

/* addEdge: TrueDistanceSelector edge-selectors.cpp:19-29, PerpendicularDistanceSelector :129-160, MultiDistanceSelector :174-227.
 * The EdgeCache relevance tests are a pure optimisation (a fresh dummy cache always passes). */
static void selector_add_edge(selector *s, const orc_shape *shape, int prevIdx, int curIdx, int nextIdx) {
    edge_t edge = load_edge(shape, curIdx);
    v2 p = s->p;
    if (s->kind == 1) {
        double dummy;
        sdist distance = edge_signed_distance(&edge, p, &dummy);
        if (sd_less(distance, s->minDistance))
            s->minDistance = distance;
        return;
    }
    int mask = s->kind == 2 ? 1 : edge.color&7; /* which ch[] take part */
    if (!mask)
        return; /* MultiDistanceSelector skips BLACK edges (edge-selectors.cpp:175-179) */
    edge_t prevEdge = load_edge(shape, prevIdx), nextEdge = load_edge(shape, nextIdx);
    double param;
    sdist distance = edge_signed_distance(&edge, p, &param);
    for (int i = 0; i < 3; ++i)
        if (mask&(1<<i))
            add_true(&s->ch[i], curIdx, distance, param);

    v2 ap = vsub(p, edge_point(&edge, 0));
    v2 bp = vsub(p, edge_point(&edge, 1));
    v2 aDir = vnormalize(edge_direction(&edge, 0), 1);
    v2 bDir = vnormalize(edge_direction(&edge, 1), 1);
    v2 prevDir = vnormalize(edge_direction(&prevEdge, 1), 1);
    v2 nextDir = vnormalize(edge_direction(&nextEdge, 0), 1);
    double add = dot(ap, vnormalize(vadd(prevDir, aDir), 1));
    double bdd = -dot(bp, vnormalize(vadd(bDir, nextDir), 1));
    if (add > 0) {
        double pd = distance.distance;
        if (get_perpendicular_distance(&pd, ap, vneg(aDir))) {
            pd = -pd;
            for (int i = 0; i < 3; ++i)
                if (mask&(1<<i))
                    add_perp(&s->ch[i], pd);
        }
    }
    if (bdd > 0) {
        double pd = distance.distance;
        if (get_perpendicular_distance(&pd, bp, bDir)) {
            for (int i = 0; i < 3; ++i)
                if (mask&(1<<i))
                    add_perp(&s->ch[i], pd);
        }
    }
}

static void selector_merge(selector *s, const selector *o) {                    /* edge-selectors.cpp:31-34, 229-233 */
    if (s->kind == 1) {
        if (sd_less(o->minDistance, s->minDistance))
            s->minDistance = o->minDistance;
        return;
    }
    for (int i = 0; i < (s->kind == 2 ? 1 : 3); ++i)
        perp_base_merge(&s->ch[i], &o->ch[i]);
}

typedef struct { double v[4]; } dist_t; /* double | MultiDistance | MultiAndTrueDistance */

static dist_t selector_distance(const selector *s, const orc_shape *shape) {    /* edge-selectors.cpp:36, 162, 235-260 */
    dist_t d = { { 0, 0, 0, 0 } };
    switch (s->kind) {
        case 1:
            d.v[0] = s->minDistance.distance;
            break;
        case 2:
            d.v[0] = perp_base_compute(shape, &s->ch[0], s->p);
            break;
        default:
            d.v[0] = perp_base_compute(shape, &s->ch[0], s->p);
            d.v[1] = perp_base_compute(shape, &s->ch[1], s->p);
            d.v[2] = perp_base_compute(shape, &s->ch[2], s->p);
            if (s->kind == 4) {                                                 /* trueDistance(), :243-250 */
                sdist t = s->ch[0].minTrue;
                if (sd_less(s->ch[1].minTrue, t))
                    t = s->ch[1].minTrue;
                if (sd_less(s->ch[2].minTrue, t))
                    t = s->ch[2].minTrue;
                d.v[3] = t.distance;
            }
    }
    return d;
}

/* ---------------------------------------------------------------------------------------- contour-combiners.cpp */

static double resolve(int kind, const dist_t *d) {                              /* contour-combiners.cpp:26-32 */
    return kind >= 3 ? dmedian(d->v[0], d->v[1], d->v[2]) : d->v[0];
}

static dist_t init_distance(int kind) {                                         /* contour-combiners.cpp:9-24 */
    dist_t d = { { 0, 0, 0, 0 } };
    int n = kind <= 2 ? 1 : kind;
    for (int i = 0; i < n; ++i)
        d.v[i] = -DBL_MAX;
    return d;
}

/* Feeds the edges of contour c in the order of ShapeDistanceFinder.hpp:45-57: cur = last, first, ..., last-1. */
static void feed_contour(selector *sel, const orc_shape *shape, int c) {
    int begin = shape->contour_offsets[c], end = shape->contour_offsets[c+1];
    int n = end-begin;
    if (n <= 0)
        return;
    int prev = n >= 2 ? end-2 : begin;
    int cur = end-1;
    for (int next = begin; next < end; ++next) {
        selector_add_edge(sel, shape, prev, cur, next);
        prev = cur;
        cur = next;
    }
}

static dist_t shape_distance(const orc_shape *shape, const int32_t *windings, selector *scratch, int kind, int overlap, v2 p) {
    int C = shape->n_contours;
    if (!overlap) {                                                             /* SimpleContourCombiner, contour-combiners.cpp:34-50 */
        selector sel;
        selector_init(&sel, kind, p);
        for (int c = 0; c < C; ++c)
            feed_contour(&sel, shape, c);
        return selector_distance(&sel, shape);
    }
    /* OverlappingContourCombiner::distance, contour-combiners.cpp:77-134 */
    selector *edgeSelectors = scratch;
    for (int c = 0; c < C; ++c) {
        selector_init(&edgeSelectors[c], kind, p);
        feed_contour(&edgeSelectors[c], shape, c);
    }
    selector shapeSel, innerSel, outerSel;
    selector_init(&shapeSel, kind, p);
    selector_init(&innerSel, kind, p);
    selector_init(&outerSel, kind, p);
    for (int i = 0; i < C; ++i) {
        dist_t edgeDistance = selector_distance(&edgeSelectors[i], shape);
        selector_merge(&shapeSel, &edgeSelectors[i]);
        if (windings[i] > 0 && resolve(kind, &edgeDistance) >= 0)
            selector_merge(&innerSel, &edgeSelectors[i]);
        if (windings[i] < 0 && resolve(kind, &edgeDistance) <= 0)
            selector_merge(&outerSel, &edgeSelectors[i]);
    }

    dist_t shapeDistance = selector_distance(&shapeSel, shape);
    dist_t innerDistance = selector_distance(&innerSel, shape);
    dist_t outerDistance = selector_distance(&outerSel, shape);
    double innerScalar = resolve(kind, &innerDistance);
    double outerScalar = resolve(kind, &outerDistance);
    dist_t distance = init_distance(kind);

    int winding = 0;
    if (innerScalar >= 0 && fabs(innerScalar) <= fabs(outerScalar)) {
        distance = innerDistance;
        winding = 1;
        for (int i = 0; i < C; ++i)
            if (windings[i] > 0) {
                dist_t contourDistance = selector_distance(&edgeSelectors[i], shape);
                if (fabs(resolve(kind, &contourDistance)) < fabs(outerScalar) && resolve(kind, &contourDistance) > resolve(kind, &distance))
                    distance = contourDistance;
            }
    } else if (outerScalar <= 0 && fabs(outerScalar) < fabs(innerScalar)) {
        distance = outerDistance;
        winding = -1;
        for (int i = 0; i < C; ++i)
            if (windings[i] < 0) {
                dist_t contourDistance = selector_distance(&edgeSelectors[i], shape);
                if (fabs(resolve(kind, &contourDistance)) < fabs(innerScalar) && resolve(kind, &contourDistance) < resolve(kind, &distance))
                    distance = contourDistance;
            }
    } else
        return shapeDistance;

    for (int i = 0; i < C; ++i)
        if (windings[i] != winding) {
            dist_t contourDistance = selector_distance(&edgeSelectors[i], shape);
            if (resolve(kind, &contourDistance)*resolve(kind, &distance) >= 0 && fabs(resolve(kind, &contourDistance)) < fabs(resolve(kind, &distance)))
                distance = contourDistance;
        }
    if (resolve(kind, &distance) == resolve(kind, &shapeDistance))
        distance = shapeDistance;
    return distance;
}

typedef struct {
    const orc_shape *shape;
    int32_t *windings;
    selector *scratch;
} finder;

static void finder_init(finder *f, const orc_shape *shape) {
    f->shape = shape;
    f->windings = (int32_t *) malloc(sizeof(int32_t)*(size_t) (shape->n_contours+1));
    f->scratch = (selector *) malloc(sizeof(selector)*(size_t) (shape->n_contours+1));
    orc_contour_windings(shape, f->windings);
}

static void finder_free(finder *f) {
    free(f->windings);
    free(f->scratch);
}

void orc_shape_distance(const orc_shape *shape, int sel, int overlap, int n, const double *pts, double *out) {
    finder f;
    finder_init(&f, shape);
    for (int i = 0; i < n; ++i) {
        dist_t d = shape_distance(shape, f.windings, f.scratch, sel, overlap, V(pts[2*i], pts[2*i+1]));
        memcpy(out+4*i, d.v, sizeof(d.v));
    }
    finder_free(&f);
}

/* ------------------------------------------------------------------------- Projection.cpp / DistanceMapping.cpp */

typedef struct {
    v2 scale, translate;            /* Projection.h:31-33 */
    double mapScale, mapTranslate;  /* DistanceMapping.h:28-30 */
} xform;

static xform make_xform(const double *xf) {
    xform t;
    t.scale = V(xf[0], xf[1]);
    t.translate = V(xf[2], xf[3]);
    t.mapScale = 1/(xf[5]-xf[4]);                                               /* DistanceMapping.cpp:13 */
    t.mapTranslate = -xf[4];
    return t;
}

static inline v2 project(const xform *t, v2 c) { return vmulv(t->scale, vadd(c, t->translate)); }      /* Projection.cpp:10-12 */
static inline v2 unproject(const xform *t, v2 c) { return vsub(vdivv(c, t->scale), t->translate); }     /* Projection.cpp:14-16 */
static inline v2 unprojectVector(const xform *t, v2 v) { return vdivv(v, t->scale); }                    /* Projection.cpp:22-24 */
static inline double map_distance(const xform *t, double d) { return t->mapScale*(d+t->mapTranslate); } /* DistanceMapping.cpp:15-17 */
static inline double map_delta(const xform *t, double d) { return t->mapScale*d; }                       /* DistanceMapping.cpp:19-21 */

/* --------------------------------------------------------------------------------------------- BitmapRef.hpp */

typedef struct {
    float *pixels;
    int width, height, rowStride, yDown, N;
} fsection;

typedef struct {
    uint8_t *pixels;
    int width, height, rowStride, yDown;
} bsection;

static inline float *fpx(const fsection *s, int x, int y) { return s->pixels+(ptrdiff_t) s->rowStride*y+s->N*x; } /* BitmapRef.hpp:88-90 */
static inline uint8_t *bpx(const bsection *s, int x, int y) { return s->pixels+(ptrdiff_t) s->rowStride*y+x; }

static void freorient(fsection *s, int yDown) {                                 /* BitmapRef.hpp:103-109 */
    if (s->yDown != yDown) {
        s->pixels += (ptrdiff_t) s->rowStride*(s->height-1);
        s->rowStride = -s->rowStride;
        s->yDown = yDown;
    }
}

static void breorient(bsection *s, int yDown) {
    if (s->yDown != yDown) {
        s->pixels += (ptrdiff_t) s->rowStride*(s->height-1);
        s->rowStride = -s->rowStride;
        s->yDown = yDown;
    }
}

/* ------------------------------------------------------------------------------------------------ msdfgen.cpp */

/* generateDistanceField<CC>, msdfgen.cpp:52-76 (row order is irrelevant to the result). */
static void generate_distance_field(const orc_shape *shape, int mode, fsection output, const xform *t, int overlap) {
    finder f;
    finder_init(&f, shape);
    freorient(&output, shape->inverse_y);                                       /* msdfgen.cpp:55 */
    for (int y = 0; y < output.height; ++y)
        for (int x = 0; x < output.width; ++x) {
            v2 p = unproject(t, V(x+.5, y+.5));                                 /* msdfgen.cpp:68 */
            dist_t d = shape_distance(shape, f.windings, f.scratch, mode, overlap, p);
            float *px = fpx(&output, x, y);
            for (int i = 0; i < output.N; ++i)
                px[i] = (float) map_distance(t, d.v[i]);                        /* msdfgen.cpp:20-48 */
        }
    finder_free(&f);
}

/* ------------------------------------------------------------------------------------ MSDFErrorCorrection.cpp */

#define ARTIFACT_T_EPSILON .01            /* MSDFErrorCorrection.cpp:16 */
#define PROTECTION_RADIUS_TOLERANCE 1.001 /* MSDFErrorCorrection.cpp:17 */
#define CLASSIFIER_FLAG_CANDIDATE 0x01
#define CLASSIFIER_FLAG_ARTIFACT 0x02
#define EC_ERROR 1                        /* MSDFErrorCorrection.h:17 */
#define EC_PROTECTED 2                    /* MSDFErrorCorrection.h:19 */

typedef struct {
    bsection stencil;
    xform t;
    double minDeviationRatio, minImproveRatio;
} ec_state;

/* Classifier: the base one (span, protectedFlag) plus, when checker != NULL, the ShapeDistanceChecker of :51-102. */
typedef struct {
    const orc_shape *shape;
    finder *f;
    int overlap;
    fsection sdf;
    const xform *t;
    v2 texelSize;
    double minImproveRatio;
    v2 shapeCoord, sdfCoord;
    const float *msd;
    int protectedFlag;
} checker;

typedef struct {
    double span;
    int protectedFlag;
    checker *parent; /* NULL -> BaseArtifactClassifier */
    v2 direction;
} classifier;

static int range_test(const classifier *c, double at, double bt, double xt, float am, float bm, float xm) { /* :30-40 */
    if ((am > .5f && bm > .5f && xm <= .5f) || (am < .5f && bm < .5f && xm >= .5f) || (!c->protectedFlag && fmedian(am, bm, xm) != xm)) {
        double axSpan = (xt-at)*c->span, bxSpan = (bt-xt)*c->span;
        if (!(xm >= am-axSpan && xm <= am+axSpan && xm >= bm-bxSpan && xm <= bm+bxSpan))
            return CLASSIFIER_FLAG_CANDIDATE|CLASSIFIER_FLAG_ARTIFACT;
        return CLASSIFIER_FLAG_CANDIDATE;
    }
    return 0;
}

static inline double dclamp(double n, double b) { return n >= 0 && n <= b ? n : (double) (n > 0)*b; }   /* arithmetics.hpp:41-43 */
static inline int iclamp(int n, int b) { return n >= 0 && n <= b ? n : (int) (n > 0)*b; }

static void interpolate(float *output, const fsection *bitmap, v2 pos) {        /* bitmap-interpolation.hpp:10-25 */
    pos.x = dclamp(pos.x, (double) bitmap->width);
    pos.y = dclamp(pos.y, (double) bitmap->height);
    pos.x -= .5, pos.y -= .5;
    int l = (int) floor(pos.x);
    int b = (int) floor(pos.y);
    int r = l+1;
    int t = b+1;
    double lr = pos.x-l;
    double bt = pos.y-b;
    l = iclamp(l, bitmap->width-1), r = iclamp(r, bitmap->width-1);
    b = iclamp(b, bitmap->height-1), t = iclamp(t, bitmap->height-1);
    for (int i = 0; i < bitmap->N; ++i)
        output[i] = fmix(fmix(fpx(bitmap, l, b)[i], fpx(bitmap, r, b)[i], lr), fmix(fpx(bitmap, l, t)[i], fpx(bitmap, r, t)[i], lr), bt);
}

static int classifier_evaluate(const classifier *c, double t, float m, int flags) { /* :42-44 and :58-82 */
    (void) m;
    if (!c->parent)
        return (flags&2) != 0;
    checker *parent = c->parent;
    if (flags&CLASSIFIER_FLAG_CANDIDATE) {
        if (flags&CLASSIFIER_FLAG_ARTIFACT)
            return 1;
        v2 tVector = smul(t, c->direction);
        float oldMSD[4], newMSD[3];
        v2 sdfCoord = vadd(parent->sdfCoord, tVector);
        interpolate(oldMSD, &parent->sdf, sdfCoord);
        double aWeight = (1-fabs(tVector.x))*(1-fabs(tVector.y));
        float aPSD = fmedian(parent->msd[0], parent->msd[1], parent->msd[2]);
        newMSD[0] = (float) (oldMSD[0]+aWeight*(aPSD-parent->msd[0]));
        newMSD[1] = (float) (oldMSD[1]+aWeight*(aPSD-parent->msd[1]));
        newMSD[2] = (float) (oldMSD[2]+aWeight*(aPSD-parent->msd[2]));
        float oldPSD = fmedian(oldMSD[0], oldMSD[1], oldMSD[2]);
        float newPSD = fmedian(newMSD[0], newMSD[1], newMSD[2]);
        dist_t ref = shape_distance(parent->shape, parent->f->windings, parent->f->scratch, 2, parent->overlap,
                                    vadd(parent->shapeCoord, vmulv(tVector, parent->texelSize)));
        float refPSD = (float) map_distance(parent->t, ref.v[0]);
        return parent->minImproveRatio*fabsf(newPSD-refPSD) < (double) fabsf(oldPSD-refPSD);
    }
    return 0;
}

static void protect_corners(ec_state *ec, const orc_shape *shape) {             /* MSDFErrorCorrection.cpp:121-151 */
    breorient(&ec->stencil, shape->inverse_y);
    bsection *st = &ec->stencil;
    for (int c = 0; c < shape->n_contours; ++c) {
        int begin = shape->contour_offsets[c], end = shape->contour_offsets[c+1];
        if (end <= begin)
            continue;
        int prevColor = shape->colors[end-1];
        for (int e = begin; e < end; ++e) {
            int commonColor = prevColor&shape->colors[e];
            if (!(commonColor&(commonColor-1))) {
                edge_t edge = load_edge(shape, e);
                v2 p = project(&ec->t, edge_point(&edge, 0));
                int l = (int) floor(p.x-.5);
                int b = (int) floor(p.y-.5);
                int r = l+1;
                int t = b+1;
                if (l < st->width && b < st->height && r >= 0 && t >= 0) {
                    if (l >= 0 && b >= 0)
                        *bpx(st, l, b) |= EC_PROTECTED;
                    if (r < st->width && b >= 0)
                        *bpx(st, r, b) |= EC_PROTECTED;
                    if (l >= 0 && t < st->height)
                        *bpx(st, l, t) |= EC_PROTECTED;
                    if (r < st->width && t < st->height)
                        *bpx(st, r, t) |= EC_PROTECTED;
                }
            }
            prevColor = shape->colors[e];
        }
    }
}

static int edge_between_texels_channel(const float *a, const float *b, int channel) { /* :154-168 */
    double t = (a[channel]-.5)/(a[channel]-b[channel]);
    if (t > 0 && t < 1) {
        float c[3] = {
            fmix(a[0], b[0], t),
            fmix(a[1], b[1], t),
            fmix(a[2], b[2], t)
        };
        return fmedian(c[0], c[1], c[2]) == c[channel];
    }
    return 0;
}

static int edge_between_texels(const float *a, const float *b) {                /* :171-177 */
    return 1*edge_between_texels_channel(a, b, 0)+2*edge_between_texels_channel(a, b, 1)+4*edge_between_texels_channel(a, b, 2);
}

static void protect_extreme_channels(uint8_t *stencil, const float *msd, float m, int mask) { /* :180-187 */
    if ((mask&1 && msd[0] != m) || (mask&2 && msd[1] != m) || (mask&4 && msd[2] != m))
        *stencil |= EC_PROTECTED;
}

static void protect_edges(ec_state *ec, const fsection *sdf) {                  /* :189-250 */
    float radius;
    bsection *st = &ec->stencil;
    int N = sdf->N;
    breorient(st, sdf->yDown);
    radius = (float) (PROTECTION_RADIUS_TOLERANCE*vlen(unprojectVector(&ec->t, V(map_delta(&ec->t, 1), 0))));
    for (int y = 0; y < sdf->height; ++y) {
        const float *left = fpx(sdf, 0, y);
        const float *right = fpx(sdf, 1, y);
        for (int x = 0; x < sdf->width-1; ++x) {
            float lm = fmedian(left[0], left[1], left[2]);
            float rm = fmedian(right[0], right[1], right[2]);
            if (fabsf(lm-.5f)+fabsf(rm-.5f) < radius) {
                int mask = edge_between_texels(left, right);
                protect_extreme_channels(bpx(st, x, y), left, lm, mask);
                protect_extreme_channels(bpx(st, x+1, y), right, rm, mask);
            }
            left += N, right += N;
        }
    }
    radius = (float) (PROTECTION_RADIUS_TOLERANCE*vlen(unprojectVector(&ec->t, V(0, map_delta(&ec->t, 1)))));
    for (int y = 0; y < sdf->height-1; ++y) {
        const float *bottom = fpx(sdf, 0, y);
        const float *top = fpx(sdf, 0, y+1);
        for (int x = 0; x < sdf->width; ++x) {
            float bm = fmedian(bottom[0], bottom[1], bottom[2]);
            float tm = fmedian(top[0], top[1], top[2]);
            if (fabsf(bm-.5f)+fabsf(tm-.5f) < radius) {
                int mask = edge_between_texels(bottom, top);
                protect_extreme_channels(bpx(st, x, y), bottom, bm, mask);
                protect_extreme_channels(bpx(st, x, y+1), top, tm, mask);
            }
            bottom += N, top += N;
        }
    }
    radius = (float) (PROTECTION_RADIUS_TOLERANCE*vlen(unprojectVector(&ec->t, V(map_delta(&ec->t, 1), map_delta(&ec->t, 1)))));
    for (int y = 0; y < sdf->height-1; ++y) {
        const float *lb = fpx(sdf, 0, y);
        const float *rb = fpx(sdf, 1, y);
        const float *lt = fpx(sdf, 0, y+1);
        const float *rt = fpx(sdf, 1, y+1);
        for (int x = 0; x < sdf->width-1; ++x) {
            float mlb = fmedian(lb[0], lb[1], lb[2]);
            float mrb = fmedian(rb[0], rb[1], rb[2]);
            float mlt = fmedian(lt[0], lt[1], lt[2]);
            float mrt = fmedian(rt[0], rt[1], rt[2]);
            if (fabsf(mlb-.5f)+fabsf(mrt-.5f) < radius) {
                int mask = edge_between_texels(lb, rt);
                protect_extreme_channels(bpx(st, x, y), lb, mlb, mask);
                protect_extreme_channels(bpx(st, x+1, y+1), rt, mrt, mask);
            }
            if (fabsf(mrb-.5f)+fabsf(mlt-.5f) < radius) {
                int mask = edge_between_texels(rb, lt);
                protect_extreme_channels(bpx(st, x+1, y), rb, mrb, mask);
                protect_extreme_channels(bpx(st, x, y+1), lt, mlt, mask);
            }
            lb += N, rb += N, lt += N, rt += N;
        }
    }
}

static void protect_all(ec_state *ec) {                                         /* :252-258 */
    for (int y = 0; y < ec->stencil.height; ++y) {
        uint8_t *mask = bpx(&ec->stencil, 0, y);
        for (int x = 0; x < ec->stencil.width; ++x)
            *mask++ |= EC_PROTECTED;
    }
}

static float interpolated_median_lin(const float *a, const float *b, double t) { /* :260-266 */
    return fmedian(fmix(a[0], b[0], t), fmix(a[1], b[1], t), fmix(a[2], b[2], t));
}

static float interpolated_median_quad(const float *a, const float *l, const float *q, double t) { /* :268-275 */
    return (float) dmedian(t*(t*q[0]+l[0])+a[0], t*(t*q[1]+l[1])+a[1], t*(t*q[2]+l[2])+a[2]);
}

static int has_linear_artifact_inner(const classifier *cl, float am, float bm, const float *a, const float *b, float dA, float dB) { /* :278-288 */
    double t = (double) dA/(dA-dB);
    if (t > ARTIFACT_T_EPSILON && t < 1-ARTIFACT_T_EPSILON) {
        float xm = interpolated_median_lin(a, b, t);
        return classifier_evaluate(cl, t, xm, range_test(cl, 0, 1, t, am, bm, xm));
    }
    return 0;
}

static int has_diagonal_artifact_inner(const classifier *cl, float am, float dm, const float *a, const float *l, const float *q,
                                       float dA, float dBC, float dD, double tEx0, double tEx1) { /* :291-327 */
    double t[2];
    int solutions = orc_solve_quadratic(t, dD-dBC+dA, dBC-dA-dA, dA);
    for (int i = 0; i < solutions; ++i) {
        if (t[i] > ARTIFACT_T_EPSILON && t[i] < 1-ARTIFACT_T_EPSILON) {
            float xm = interpolated_median_quad(a, l, q, t[i]);
            int rangeFlags = range_test(cl, 0, 1, t[i], am, dm, xm);
            double tEnd[2];
            float em[2];
            if (tEx0 > 0 && tEx0 < 1) {
                tEnd[0] = 0, tEnd[1] = 1;
                em[0] = am, em[1] = dm;
                tEnd[tEx0 > t[i]] = tEx0;
                em[tEx0 > t[i]] = interpolated_median_quad(a, l, q, tEx0);
                rangeFlags |= range_test(cl, tEnd[0], tEnd[1], t[i], em[0], em[1], xm);
            }
            if (tEx1 > 0 && tEx1 < 1) {
                tEnd[0] = 0, tEnd[1] = 1;
                em[0] = am, em[1] = dm;
                tEnd[tEx1 > t[i]] = tEx1;
                em[tEx1 > t[i]] = interpolated_median_quad(a, l, q, tEx1);
                rangeFlags |= range_test(cl, tEnd[0], tEnd[1], t[i], em[0], em[1], xm);
            }
            if (classifier_evaluate(cl, t[i], xm, rangeFlags))
                return 1;
        }
    }
    return 0;
}

static int has_linear_artifact(const classifier *cl, float am, const float *a, const float *b) { /* :330-342 */
    float bm = fmedian(b[0], b[1], b[2]);
    return (
        fabsf(am-.5f) >= fabsf(bm-.5f) && (
            has_linear_artifact_inner(cl, am, bm, a, b, a[1]-a[0], b[1]-b[0]) ||
            has_linear_artifact_inner(cl, am, bm, a, b, a[2]-a[1], b[2]-b[1]) ||
            has_linear_artifact_inner(cl, am, bm, a, b, a[0]-a[2], b[0]-b[2])
        )
    );
}

static int has_diagonal_artifact(const classifier *cl, float am, const float *a, const float *b, const float *c, const float *d) { /* :345-381 */
    float dm = fmedian(d[0], d[1], d[2]);
    if (fabsf(am-.5f) >= fabsf(dm-.5f)) {
        float abc[3] = { a[0]-b[0]-c[0], a[1]-b[1]-c[1], a[2]-b[2]-c[2] };
        float l[3] = { -a[0]-abc[0], -a[1]-abc[1], -a[2]-abc[2] };
        float q[3] = { d[0]+abc[0], d[1]+abc[1], d[2]+abc[2] };
        double tEx[3] = { -.5*l[0]/q[0], -.5*l[1]/q[1], -.5*l[2]/q[2] };
        return (
            has_diagonal_artifact_inner(cl, am, dm, a, l, q, a[1]-a[0], b[1]-b[0]+c[1]-c[0], d[1]-d[0], tEx[0], tEx[1]) ||
            has_diagonal_artifact_inner(cl, am, dm, a, l, q, a[2]-a[1], b[2]-b[1]+c[2]-c[1], d[2]-d[1], tEx[1], tEx[2]) ||
            has_diagonal_artifact_inner(cl, am, dm, a, l, q, a[0]-a[2], b[0]-b[2]+c[0]-c[2], d[0]-d[2], tEx[2], tEx[0])
        );
    }
    return 0;
}

static void spans(const ec_state *ec, double *hSpan, double *vSpan, double *dSpan) { /* :386-389, :416-419 */
    *hSpan = ec->minDeviationRatio*vlen(unprojectVector(&ec->t, V(map_delta(&ec->t, 1), 0)));
    *vSpan = ec->minDeviationRatio*vlen(unprojectVector(&ec->t, V(0, map_delta(&ec->t, 1))));
    *dSpan = ec->minDeviationRatio*vlen(unprojectVector(&ec->t, V(map_delta(&ec->t, 1), map_delta(&ec->t, 1))));
}

static classifier mk_classifier(checker *parent, double dx, double dy, double span, int protectedFlag) {
    classifier c;
    c.span = span;
    c.protectedFlag = protectedFlag;
    c.parent = parent;
    c.direction = V(dx, dy);
    return c;
}

/* Shared body of findErrors<N>(sdf) (:383-410, chk == NULL) and findErrors<CC,N>(sdf, shape) (:412-457). */
static int texel_has_error(const fsection *sdf, int x, int y, checker *chk, int protectedFlag, double hSpan, double vSpan, double dSpan) {
    const float *c = fpx(sdf, x, y);
    float cm = fmedian(c[0], c[1], c[2]);
    const float *l = NULL, *b = NULL, *r = NULL, *t = NULL;
    classifier cl;
    int w = sdf->width, h = sdf->height;
    return (
        (x > 0 && ((l = fpx(sdf, x-1, y)), (cl = mk_classifier(chk, -1, 0, hSpan, protectedFlag)), has_linear_artifact(&cl, cm, c, l))) ||
        (y > 0 && ((b = fpx(sdf, x, y-1)), (cl = mk_classifier(chk, 0, -1, vSpan, protectedFlag)), has_linear_artifact(&cl, cm, c, b))) ||
        (x < w-1 && ((r = fpx(sdf, x+1, y)), (cl = mk_classifier(chk, +1, 0, hSpan, protectedFlag)), has_linear_artifact(&cl, cm, c, r))) ||
        (y < h-1 && ((t = fpx(sdf, x, y+1)), (cl = mk_classifier(chk, 0, +1, vSpan, protectedFlag)), has_linear_artifact(&cl, cm, c, t))) ||
        (x > 0 && y > 0 && ((cl = mk_classifier(chk, -1, -1, dSpan, protectedFlag)), has_diagonal_artifact(&cl, cm, c, l, b, fpx(sdf, x-1, y-1)))) ||
        (x < w-1 && y > 0 && ((cl = mk_classifier(chk, +1, -1, dSpan, protectedFlag)), has_diagonal_artifact(&cl, cm, c, r, b, fpx(sdf, x+1, y-1)))) ||
        (x > 0 && y < h-1 && ((cl = mk_classifier(chk, -1, +1, dSpan, protectedFlag)), has_diagonal_artifact(&cl, cm, c, l, t, fpx(sdf, x-1, y+1)))) ||
        (x < w-1 && y < h-1 && ((cl = mk_classifier(chk, +1, +1, dSpan, protectedFlag)), has_diagonal_artifact(&cl, cm, c, r, t, fpx(sdf, x+1, y+1))))
    );
}

static void find_errors_sdf(ec_state *ec, const fsection *sdf) {                 /* :383-410 */
    breorient(&ec->stencil, sdf->yDown);
    double hSpan, vSpan, dSpan;
    spans(ec, &hSpan, &vSpan, &dSpan);
    for (int y = 0; y < sdf->height; ++y)
        for (int x = 0; x < sdf->width; ++x) {
            uint8_t *s = bpx(&ec->stencil, x, y);
            int protectedFlag = (*s&EC_PROTECTED) != 0;
            *s |= (uint8_t) (EC_ERROR*texel_has_error(sdf, x, y, NULL, protectedFlag, hSpan, vSpan, dSpan));
        }
}

static void find_errors_shape(ec_state *ec, fsection sdf, const orc_shape *shape, int overlap) { /* :412-457 */
    freorient(&sdf, shape->inverse_y);
    breorient(&ec->stencil, sdf.yDown);
    double hSpan, vSpan, dSpan;
    spans(ec, &hSpan, &vSpan, &dSpan);
    finder f;
    finder_init(&f, shape);
    checker chk;
    chk.shape = shape;
    chk.f = &f;
    chk.overlap = overlap;
    chk.sdf = sdf;
    chk.t = &ec->t;
    chk.texelSize = unprojectVector(&ec->t, V(1, 1));                           /* :90 */
    chk.minImproveRatio = ec->minImproveRatio;
    for (int y = 0; y < sdf.height; ++y)
        for (int x = 0; x < sdf.width; ++x) {
            uint8_t *s = bpx(&ec->stencil, x, y);
            if (*s&EC_ERROR)
                continue;
            chk.shapeCoord = unproject(&ec->t, V(x+.5, y+.5));
            chk.sdfCoord = V(x+.5, y+.5);
            chk.msd = fpx(&sdf, x, y);
            chk.protectedFlag = (*s&EC_PROTECTED) != 0;
            *s |= (uint8_t) (EC_ERROR*texel_has_error(&sdf, x, y, &chk, chk.protectedFlag, hSpan, vSpan, dSpan));
        }
    finder_free(&f);
}

static void ec_apply(const ec_state *ec, fsection sdf) {                        /* :459-479 */
    freorient(&sdf, ec->stencil.yDown);
    for (int y = 0; y < sdf.height; ++y)
        for (int x = 0; x < sdf.width; ++x)
            if (*bpx(&ec->stencil, x, y)&EC_ERROR) {
                float *pixel = fpx(&sdf, x, y);
                float m = fmedian(pixel[0], pixel[1], pixel[2]);
                pixel[0] = m, pixel[1] = m, pixel[2] = m;
            }
}

/* ----------------------------------------------------------------------------------- msdf-error-correction.cpp */

enum { EC_DISABLED = 0, EC_INDISCRIMINATE = 1, EC_EDGE_PRIORITY = 2, EC_EDGE_ONLY = 3 };               /* generator-config.h:22-31 */
enum { EC_DO_NOT_CHECK_DISTANCE = 0, EC_CHECK_DISTANCE_AT_EDGE = 1, EC_ALWAYS_CHECK_DISTANCE = 2 };    /* generator-config.h:33-40 */

static void ec_state_init(ec_state *ec, uint8_t *buffer, int w, int h, const xform *t, double minDev, double minImp) {
    ec->stencil.pixels = buffer;
    ec->stencil.width = w, ec->stencil.height = h, ec->stencil.rowStride = w, ec->stencil.yDown = 0;
    ec->t = *t;
    ec->minDeviationRatio = minDev;
    ec->minImproveRatio = minImp;
    memset(buffer, 0, (size_t) w*h);                                            /* MSDFErrorCorrection.cpp:109-110 */
}

static void error_correction_inner(const fsection *sdf, const orc_shape *shape, const xform *t, int overlap, int ecMode, int ecDist,
                                   double minDev, double minImp, uint8_t *buffer) { /* msdf-error-correction.cpp:12-48 */
    if (ecMode == EC_DISABLED)
        return;
    uint8_t *own = NULL;
    if (!buffer)
        buffer = own = (uint8_t *) malloc((size_t) sdf->width*sdf->height+1);
    ec_state ec;
    ec_state_init(&ec, buffer, sdf->width, sdf->height, t, minDev, minImp);
    switch (ecMode) {
        case EC_EDGE_PRIORITY:
            protect_corners(&ec, shape);
            protect_edges(&ec, sdf);
            break;
        case EC_EDGE_ONLY:
            protect_all(&ec);
            break;
        default:
            break;
    }
    if (ecDist == EC_DO_NOT_CHECK_DISTANCE || (ecDist == EC_CHECK_DISTANCE_AT_EDGE && ecMode != EC_EDGE_ONLY)) {
        find_errors_sdf(&ec, sdf);
        if (ecDist == EC_CHECK_DISTANCE_AT_EDGE)
            protect_all(&ec);
    }
    if (ecDist == EC_ALWAYS_CHECK_DISTANCE || ecDist == EC_CHECK_DISTANCE_AT_EDGE)
        find_errors_shape(&ec, *sdf, shape, overlap);
    ec_apply(&ec, *sdf);
    free(own);
}

void orc_generate(const orc_shape *shape, int mode, float *pixels, int w, int h, int row_stride, int y_down, const double *xf,
                  int overlap, int ec_mode, int ec_dist, double min_dev, double min_imp, uint8_t *stencil) {
    xform t = make_xform(xf);
    fsection out;
    out.pixels = pixels, out.width = w, out.height = h, out.rowStride = row_stride, out.yDown = y_down;
    out.N = mode <= 2 ? 1 : mode;
    generate_distance_field(shape, mode, out, &t, overlap);                     /* msdfgen.cpp:78-106 */
    if (mode >= 3)
        error_correction_inner(&out, shape, &t, overlap, ec_mode, ec_dist, min_dev, min_imp, stencil);
}

void orc_error_correction(const orc_shape *shape, int channels, float *pixels, int w, int h, int row_stride, int y_down, const double *xf,
                          int overlap, int ec_mode, int ec_dist, double min_dev, double min_imp, uint8_t *stencil) {
    xform t = make_xform(xf);
    fsection sdf;
    sdf.pixels = pixels, sdf.width = w, sdf.height = h, sdf.rowStride = row_stride, sdf.yDown = y_down, sdf.N = channels;
    error_correction_inner(&sdf, shape, &t, overlap, ec_mode, ec_dist, min_dev, min_imp, stencil);
}

void orc_ec_stages(const orc_shape *shape, int channels, const float *pixels, int w, int h, const double *xf, int overlap,
                   double min_dev, double min_imp, uint8_t *stages) {
    xform t = make_xform(xf);
    size_t n = (size_t) w*h;
    uint8_t *buf = (uint8_t *) malloc(n+1);
    fsection sdf;
    sdf.pixels = (float *) pixels, sdf.width = w, sdf.height = h, sdf.rowStride = w*channels, sdf.yDown = 0, sdf.N = channels;
    ec_state ec;
    ec_state_init(&ec, buf, w, h, &t, min_dev, min_imp);
    protect_corners(&ec, shape);
    memcpy(stages, buf, n);
    protect_edges(&ec, &sdf);
    memcpy(stages+n, buf, n);
    find_errors_sdf(&ec, &sdf);
    memcpy(stages+2*n, buf, n);
    protect_all(&ec);
    find_errors_shape(&ec, sdf, shape, overlap);
    memcpy(stages+3*n, buf, n);
    free(buf);
}

/* ------------------------------------------------------------------------ scanline fill / distanceSignCorrection */

static int scan_linear(const v2 *p, double *x, int *dy, double y) {                /* edge-segments.cpp:279-287 */
    if ((y >= p[0].y && y < p[1].y) || (y >= p[1].y && y < p[0].y)) {
        double param = (y-p[0].y)/(p[1].y-p[0].y);
        x[0] = (1.-param)*p[0].x+param*p[1].x;                                    /* mix(p0.x, p1.x, param), arithmetics.hpp:27-31 */
        dy[0] = isign(p[1].y-p[0].y);
        return 1;
    }
    return 0;
}

static int scan_quadratic(const v2 *p, double *x, int *dy, double y) {             /* edge-segments.cpp:289-341 */
    int total = 0;
    int nextDY = y > p[0].y ? 1 : -1;
    x[total] = p[0].x;
    if (p[0].y == y) {
        if (p[0].y < p[1].y || (p[0].y == p[1].y && p[0].y < p[2].y))
            dy[total++] = 1;
        else
            nextDY = 1;
    }
    {
        v2 ab = vsub(p[1], p[0]);
        v2 br = vsub(vsub(p[2], p[1]), ab);
        double t[2];
        int solutions = orc_solve_quadratic(t, br.y, 2*ab.y, p[0].y-y);
        double tmp;
        if (solutions >= 2 && t[0] > t[1])
            tmp = t[0], t[0] = t[1], t[1] = tmp;
        for (int i = 0; i < solutions && total < 2; ++i) {
            if (t[i] >= 0 && t[i] <= 1) {
                x[total] = p[0].x+2*t[i]*ab.x+t[i]*t[i]*br.x;
                if (nextDY*(ab.y+t[i]*br.y) >= 0) {
                    dy[total++] = nextDY;
                    nextDY = -nextDY;
                }
            }
        }
    }
    if (p[2].y == y) {
        if (nextDY > 0 && total > 0) {
            --total;
            nextDY = -1;
        }
        if ((p[2].y < p[1].y || (p[2].y == p[1].y && p[2].y < p[0].y)) && total < 2) {
            x[total] = p[2].x;
            if (nextDY < 0) {
                dy[total++] = -1;
                nextDY = 1;
            }
        }
    }
    if (nextDY != (y >= p[2].y ? 1 : -1)) {
        if (total > 0)
            --total;
        else {
            if (fabs(p[2].y-y) < fabs(p[0].y-y))
                x[total] = p[2].x;
            dy[total++] = nextDY;
        }
    }
    return total;
}

static int scan_cubic(const v2 *p, double *x, int *dy, double y) {                 /* edge-segments.cpp:343-403 */
    int total = 0;
    int nextDY = y > p[0].y ? 1 : -1;
    x[total] = p[0].x;
    if (p[0].y == y) {
        if (p[0].y < p[1].y || (p[0].y == p[1].y && (p[0].y < p[2].y || (p[0].y == p[2].y && p[0].y < p[3].y))))
            dy[total++] = 1;
        else
            nextDY = 1;
    }
    {
        v2 ab = vsub(p[1], p[0]);
        v2 br = vsub(vsub(p[2], p[1]), ab);
        v2 as = vsub(vsub(vsub(p[3], p[2]), vsub(p[2], p[1])), br);
        double t[3];
        int solutions = orc_solve_cubic(t, as.y, 3*br.y, 3*ab.y, p[0].y-y);
        double tmp;
        if (solutions >= 2) {
            if (t[0] > t[1])
                tmp = t[0], t[0] = t[1], t[1] = tmp;
            if (solutions >= 3 && t[1] > t[2]) {
                tmp = t[1], t[1] = t[2], t[2] = tmp;
                if (t[0] > t[1])
                    tmp = t[0], t[0] = t[1], t[1] = tmp;
            }
        }
        for (int i = 0; i < solutions && total < 3; ++i) {
            if (t[i] >= 0 && t[i] <= 1) {
                x[total] = p[0].x+3*t[i]*ab.x+3*t[i]*t[i]*br.x+t[i]*t[i]*t[i]*as.x;
                if (nextDY*(ab.y+2*t[i]*br.y+t[i]*t[i]*as.y) >= 0) {
                    dy[total++] = nextDY;
                    nextDY = -nextDY;
                }
            }
        }
    }
    if (p[3].y == y) {
        if (nextDY > 0 && total > 0) {
            --total;
            nextDY = -1;
        }
        if ((p[3].y < p[2].y || (p[3].y == p[2].y && (p[3].y < p[1].y || (p[3].y == p[1].y && p[3].y < p[0].y)))) && total < 3) {
            x[total] = p[3].x;
            if (nextDY < 0) {
                dy[total++] = -1;
                nextDY = 1;
            }
        }
    }
    if (nextDY != (y >= p[3].y ? 1 : -1)) {
        if (total > 0)
            --total;
        else {
            if (fabs(p[3].y-y) < fabs(p[0].y-y))
                x[total] = p[3].x;
            dy[total++] = nextDY;
        }
    }
    return total;
}

static int edge_scanline(const edge_t *e, double *x, int *dy, double y) {
    switch (e->type) {
        case 1: return scan_linear(e->p, x, dy, y);
        case 2: return scan_quadratic(e->p, x, dy, y);
        default: return scan_cubic(e->p, x, dy, y);
    }
}

int orc_scanline_intersections(int type, const double *p, double y, double *x, int32_t *dy) {
    edge_t e;
    e.type = type, e.color = 7;
    for (int i = 0; i < 4; ++i)
        e.p[i] = V(p[2*i], p[2*i+1]);
    int d[3] = { 0, 0, 0 };
    int n = edge_scanline(&e, x, d, y);
    for (int i = 0; i < 3; ++i)
        dy[i] = d[i];
    return n;
}

static int interpret_fill_rule(int intersections, int rule) {                    /* Scanline.cpp:13-25 */
    switch (rule) {
        case 0: return intersections != 0;
        case 1: return intersections&1;
        case 2: return intersections > 0;
        case 3: return intersections < 0;
    }
    return 0;
}

typedef struct { double *x; int *dir; int n, cap; } scanline_t;

/* Shape::scanline (Shape.cpp:117-135). The reference sorts the intersections and prefix-sums the directions (Scanline.cpp:66-77);
 * filled(x) (Scanline.cpp:120-122) then reads the sum over all intersections with x_i <= x, which needs no ordering. */
static void shape_scanline(const orc_shape *s, scanline_t *line, double y) {
    int nE = s->contour_offsets[s->n_contours];
    line->n = 0;
    for (int e = 0; e < nE; ++e) {
        edge_t edge = load_edge(s, e);
        double x[3];
        int dy[3];
        int n = edge_scanline(&edge, x, dy, y);
        for (int i = 0; i < n; ++i) {
            line->x[line->n] = x[i];
            line->dir[line->n++] = dy[i];
        }
    }
}

static int scanline_filled(const scanline_t *line, double x, int rule) {
    int sum = 0;
    for (int i = 0; i < line->n; ++i)
        if (x >= line->x[i])
            sum += line->dir[i];
    return interpret_fill_rule(sum, rule);
}

void orc_rasterize(const orc_shape *shape, float *pixels, int w, int h, int row_stride, int y_down, const double *xf4, int fillRule) {
    fsection out;                                                                /* rasterization.cpp:8-16 */
    out.pixels = pixels, out.width = w, out.height = h, out.rowStride = row_stride, out.yDown = y_down, out.N = 1;
    freorient(&out, shape->inverse_y);
    int nE = shape->contour_offsets[shape->n_contours];
    scanline_t line;
    line.cap = 3*nE+1;
    line.x = (double *) malloc(sizeof(double)*(size_t) line.cap);
    line.dir = (int *) malloc(sizeof(int)*(size_t) line.cap);
    for (int y = 0; y < h; ++y) {
        shape_scanline(shape, &line, (y+.5)/xf4[1]-xf4[3]);
        for (int x = 0; x < w; ++x)
            *fpx(&out, x, y) = (float) scanline_filled(&line, (x+.5)/xf4[0]-xf4[2], fillRule);
    }
    free(line.x);
    free(line.dir);
}

void orc_sign_correction(const orc_shape *shape, int channels, float *pixels, int w, int h, int row_stride, int y_down, const double *xf4,
                         float sdfZeroValue, int fillRule) {                   /* rasterization.cpp:19-88 */
    if (!(w && h))
        return;
    fsection sdf;
    sdf.pixels = pixels, sdf.width = w, sdf.height = h, sdf.rowStride = row_stride, sdf.yDown = y_down, sdf.N = channels;
    freorient(&sdf, shape->inverse_y);
    xform t;
    t.scale = V(xf4[0], xf4[1]), t.translate = V(xf4[2], xf4[3]), t.mapScale = 1, t.mapTranslate = 0;
    float doubleSdfZeroValue = sdfZeroValue+sdfZeroValue;
    int nE = shape->contour_offsets[shape->n_contours];
    scanline_t line;
    line.cap = 3*nE+1;
    line.x = (double *) malloc(sizeof(double)*(size_t) line.cap);
    line.dir = (int *) malloc(sizeof(int)*(size_t) line.cap);
    char *matchMap = (char *) calloc((size_t) w*h+1, 1);
    int ambiguous = 0;
    char *match = matchMap;
    for (int y = 0; y < h; ++y) {
        shape_scanline(shape, &line, (y+.5)/t.scale.y-t.translate.y);             /* projection.unprojectY(y+.5), Projection.cpp:38-40 */
        for (int x = 0; x < w; ++x) {
            int fill = scanline_filled(&line, (x+.5)/t.scale.x-t.translate.x, fillRule);
            float *msd = fpx(&sdf, x, y);
            if (channels == 1) {                                                 /* :19-33 */
                if ((msd[0] > sdfZeroValue) != fill)
                    msd[0] = doubleSdfZeroValue-msd[0];
            } else {                                                             /* :35-88 */
                float sd = fmedian(msd[0], msd[1], msd[2]);
                if (sd == sdfZeroValue)
                    ambiguous = 1;
                else if ((sd > sdfZeroValue) != fill) {
                    msd[0] = doubleSdfZeroValue-msd[0];
                    msd[1] = doubleSdfZeroValue-msd[1];
                    msd[2] = doubleSdfZeroValue-msd[2];
                    *match = -1;
                } else
                    *match = 1;
                if (channels >= 4 && (msd[3] > sdfZeroValue) != fill)
                    msd[3] = doubleSdfZeroValue-msd[3];
            }
            ++match;
        }
    }
    if (ambiguous) {
        match = matchMap;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                if (!*match) {
                    int neighborMatch = 0;
                    if (x > 0) neighborMatch += *(match-1);
                    if (x < w-1) neighborMatch += *(match+1);
                    if (y > 0) neighborMatch += *(match-w);
                    if (y < h-1) neighborMatch += *(match+w);
                    if (neighborMatch < 0) {
                        float *msd = fpx(&sdf, x, y);
                        msd[0] = doubleSdfZeroValue-msd[0];
                        msd[1] = doubleSdfZeroValue-msd[1];
                        msd[2] = doubleSdfZeroValue-msd[2];
                    }
                }
                ++match;
            }
    }
    free(matchMap);
    free(line.x);
    free(line.dir);
}

/* ---------------------------------------------------------------------------------------- CPU baseline helper */

typedef struct {
    const orc_shape *shapes;
    int n_glyphs, mode, w, h, overlap, ec_mode, ec_dist;
    float *pixels;
    const double *xfs;
    double min_dev, min_imp;
    int *next;
    pthread_mutex_t *lock;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *job = (batch_job *) arg;
    int N = job->mode <= 2 ? 1 : job->mode;
    uint8_t *stencil = (uint8_t *) malloc((size_t) job->w*job->h+1);
    for (;;) {
        pthread_mutex_lock(job->lock);
        int g = (*job->next)++;
        pthread_mutex_unlock(job->lock);
        if (g >= job->n_glyphs)
            break;
        orc_generate(&job->shapes[g], job->mode, job->pixels+(size_t) g*job->w*job->h*N, job->w, job->h, job->w*N, 0, job->xfs+6*g,
                     job->overlap, job->ec_mode, job->ec_dist, job->min_dev, job->min_imp, stencil);
    }
    free(stencil);
    return NULL;
}

double orc_generate_batch_timed(const orc_shape *shapes, int n_glyphs, int mode, float *pixels, int w, int h, const double *xfs,
                                int overlap, int ec_mode, int ec_dist, double min_dev, double min_imp, int threads) {
    int next = 0;
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    batch_job job = { shapes, n_glyphs, mode, w, h, overlap, ec_mode, ec_dist, pixels, xfs, min_dev, min_imp, &next, &lock };
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (threads <= 1)
        batch_worker(&job);
    else {
        pthread_t *pool = (pthread_t *) malloc(sizeof(pthread_t)*(size_t) threads);
        for (int i = 0; i < threads; ++i)
            pthread_create(&pool[i], NULL, batch_worker, &job);
        for (int i = 0; i < threads; ++i)
            pthread_join(pool[i], NULL);
        free(pool);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double) (t1.tv_sec-t0.tv_sec)+1e-9*(double) (t1.tv_nsec-t0.tv_nsec);
}

/* pixelFloatToByte (core/pixel-conversion.hpp:8-10) over n floats; clamp(x) per core/arithmetics.hpp:35-37. */
void orc_pixel_float_to_byte(const float *in, unsigned char *out, long n) {
    for (long i = 0; i < n; ++i) {
        float x = in[i];
        float c = x >= 0.f && x <= 1.f ? x : (float) (x > 0.f);
        out[i] = (unsigned char) ~(int) (255.5f-255.f*c);
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * shape preparation -- Shape::normalize (core/Shape.cpp:65-92) and edgeColoringSimple
 * (core/edge-coloring.cpp:68-142) on the flat shape model. Output arrays must hold 3*E edges (worst case: every edge split in thirds).
 */
#define CORNER_DOT_EPSILON .000001                 /* core/Shape.h:12 */
#define DECONVERGE_OVERSHOOT 1.11111111111111111   /* core/Shape.cpp:7 */

static inline int veq(v2 a, v2 b) { return a.x == b.x && a.y == b.y; }                      /* Vector2.hpp:67-69 */
static inline v2 vorthogonal(v2 a, int polarity) { return polarity ? V(-a.y, a.x) : V(a.y, -a.x); } /* Vector2.hpp:49-51 */

static edge_t mk_edge(int type, int color, v2 a, v2 b, v2 c, v2 d) {
    edge_t e;
    e.type = type, e.color = color;
    e.p[0] = a, e.p[1] = b, e.p[2] = c, e.p[3] = d;
    return e;
}

static void split_in_thirds(const edge_t *e, edge_t *part) {                                 /* edge-segments.cpp:508-527 */
    const v2 *p = e->p;
    const v2 z = V(0, 0);
    switch (e->type) {
        case 1:
            part[0] = mk_edge(1, e->color, p[0], edge_point(e, 1/3.), z, z);
            part[1] = mk_edge(1, e->color, edge_point(e, 1/3.), edge_point(e, 2/3.), z, z);
            part[2] = mk_edge(1, e->color, edge_point(e, 2/3.), p[1], z, z);
            break;
        case 2:
            part[0] = mk_edge(2, e->color, p[0], vmix(p[0], p[1], 1/3.), edge_point(e, 1/3.), z);
            part[1] = mk_edge(2, e->color, edge_point(e, 1/3.), vmix(vmix(p[0], p[1], 5/9.), vmix(p[1], p[2], 4/9.), .5), edge_point(e, 2/3.), z);
            part[2] = mk_edge(2, e->color, edge_point(e, 2/3.), vmix(p[1], p[2], 2/3.), p[2], z);
            break;
        default:
            part[0] = mk_edge(3, e->color, p[0], veq(p[0], p[1]) ? p[0] : vmix(p[0], p[1], 1/3.),
                              vmix(vmix(p[0], p[1], 1/3.), vmix(p[1], p[2], 1/3.), 1/3.), edge_point(e, 1/3.));
            part[1] = mk_edge(3, e->color, edge_point(e, 1/3.),
                              vmix(vmix(vmix(p[0], p[1], 1/3.), vmix(p[1], p[2], 1/3.), 1/3.), vmix(vmix(p[1], p[2], 1/3.), vmix(p[2], p[3], 1/3.), 1/3.), 2/3.),
                              vmix(vmix(vmix(p[0], p[1], 2/3.), vmix(p[1], p[2], 2/3.), 2/3.), vmix(vmix(p[1], p[2], 2/3.), vmix(p[2], p[3], 2/3.), 2/3.), 1/3.),
                              edge_point(e, 2/3.));
            part[2] = mk_edge(3, e->color, edge_point(e, 2/3.), vmix(vmix(p[1], p[2], 2/3.), vmix(p[2], p[3], 2/3.), 2/3.),
                              veq(p[2], p[3]) ? p[3] : vmix(p[2], p[3], 2/3.), p[3]);
            break;
    }
}

static void deconverge_edge(edge_t *e, int param, v2 vector) {                               /* Shape.cpp:44-62 */
    if (e->type == 2) {                                                                      /* convertToCubic, edge-segments.cpp:529-531 */
        v2 a = e->p[0], b = e->p[1], c = e->p[2];
        *e = mk_edge(3, e->color, a, vmix(a, b, 2/3.), vmix(b, c, 1/3.), c);
    }
    if (e->type == 3) {
        if (param == 0)
            e->p[1] = vadd(e->p[1], smul(vlen(vsub(e->p[1], e->p[0])), vector));
        else
            e->p[2] = vadd(e->p[2], smul(vlen(vsub(e->p[2], e->p[3])), vector));
    }
}

static void simplify_degenerate_curve(v2 *cp, int *order) {                                  /* convergent-curve-ordering.cpp:34-46 */
    if (*order == 3 && (veq(cp[1], cp[0]) || veq(cp[1], cp[3])) && (veq(cp[2], cp[0]) || veq(cp[2], cp[3]))) {
        cp[1] = cp[3];
        *order = 1;
    }
    if (*order == 2 && (veq(cp[1], cp[0]) || veq(cp[1], cp[2]))) {
        cp[1] = cp[2];
        *order = 1;
    }
    if (*order == 1 && veq(cp[0], cp[1]))
        *order = 0;
}

static int curve_ordering_at(const v2 *corner, int before, int after) {                      /* convergent-curve-ordering.cpp:48-119 */
    if (!(before > 0 && after > 0))
        return 0;
    v2 a1, a2 = V(0, 0), a3 = V(0, 0), b1, b2 = V(0, 0), b3 = V(0, 0);
    a1 = vsub(corner[-1], corner[0]);
    b1 = vsub(corner[1], corner[0]);
    if (before >= 2)
        a2 = vsub(vsub(corner[-2], corner[-1]), a1);
    if (after >= 2)
        b2 = vsub(vsub(corner[2], corner[1]), b1);
    if (before >= 3) {
        a3 = vsub(vsub(vsub(corner[-3], corner[-2]), vsub(corner[-2], corner[-1])), a2);
        a2 = smul(3, a2);                                                                    /* Vector2 *= double: x *= v, y *= v */
    }
    if (after >= 3) {
        b3 = vsub(vsub(vsub(corner[3], corner[2]), vsub(corner[2], corner[1])), b2);
        b2 = smul(3, b2);
    }
    a1 = smul(before, a1);
    b1 = smul(after, b1);
    double d;
    if (vnonzero(a1) && vnonzero(b1)) {
        double as = vlen(a1), bs = vlen(b1);
        if ((d = as*cross(a1, b2)+bs*cross(a2, b1)))
            return isign(d);
        if ((d = as*as*cross(a1, b3)+as*bs*cross(a2, b2)+bs*bs*cross(a3, b1)))
            return isign(d);
        if ((d = as*cross(a2, b3)+bs*cross(a3, b2)))
            return isign(d);
        return isign(cross(a3, b3));
    }
    int s = 1;
    if (vnonzero(a1)) {
        b1 = a1;
        a1 = b2, b2 = a2, a2 = a1;
        a1 = b3, b3 = a3, a3 = a1;
        s = -1;
    }
    if (vnonzero(b1)) {
        if ((d = cross(a3, b1)))
            return s*isign(d);
        if ((d = cross(a2, b2)))
            return s*isign(d);
        if ((d = cross(a3, b2)))
            return s*isign(d);
        if ((d = cross(a2, b3)))
            return s*isign(d);
        return s*isign(cross(a3, b3));
    }
    if ((d = sqrt(vlen(a2))*cross(a2, b3)+sqrt(vlen(b2))*cross(a3, b2)))
        return isign(d);
    return isign(cross(a3, b3));
}

static int convergent_curve_ordering(const edge_t *a, const edge_t *b) {                     /* convergent-curve-ordering.cpp:121-138 */
    v2 cps[12];
    for (int i = 0; i < 12; ++i)
        cps[i] = V(0, 0);
    v2 *corner = cps+4, *tmp = cps+8;
    int ao = a->type, bo = b->type;
    for (int i = 0; i <= ao; ++i)
        tmp[i] = a->p[i];
    for (int i = 0; i <= bo; ++i)
        corner[i] = b->p[i];
    if (!veq(tmp[ao], corner[0]))
        return 0;
    simplify_degenerate_curve(tmp, &ao);
    simplify_degenerate_curve(corner, &bo);
    for (int i = 0; i < ao; ++i)
        corner[i-ao] = tmp[i];
    return curve_ordering_at(corner, ao, bo);
}

/* Shape::normalize on one contour; edges[] has room for 3 entries at least. Returns the new edge count. */
static int normalize_contour(edge_t *edges, int n) {
    if (n == 1) {
        edge_t parts[3];
        split_in_thirds(&edges[0], parts);
        edges[0] = parts[0], edges[1] = parts[1], edges[2] = parts[2];
        return 3;
    }
    if (n > 0) {
        int prev = n-1;
        for (int i = 0; i < n; ++i) {
            v2 prevDir = vnormalize(edge_direction(&edges[prev], 1), 0);
            v2 curDir = vnormalize(edge_direction(&edges[i], 0), 0);
            if (dot(prevDir, curDir) < CORNER_DOT_EPSILON-1) {
                double factor = DECONVERGE_OVERSHOOT*sqrt(1-(CORNER_DOT_EPSILON-1)*(CORNER_DOT_EPSILON-1))/(CORNER_DOT_EPSILON-1);
                v2 axis = smul(factor, vnormalize(vsub(curDir, prevDir), 0));
                if (convergent_curve_ordering(&edges[prev], &edges[i]) < 0)
                    axis = vneg(axis);
                deconverge_edge(&edges[prev], 1, vorthogonal(axis, 1));
                deconverge_edge(&edges[i], 0, vorthogonal(axis, 0));
            }
            prev = i;
        }
    }
    return n;
}

static int seed_extract2(unsigned long long *seed) { int v = (int) (*seed)&1; *seed >>= 1; return v; }   /* edge-coloring.cpp:36-40 */
static int seed_extract3(unsigned long long *seed) { int v = (int) (*seed%3); *seed /= 3; return v; }      /* :42-46 */
static void switch_color(int *color, unsigned long long *seed) {                                          /* :53-56 */
    int shifted = *color<<(1+seed_extract2(seed));
    *color = (shifted|shifted>>3)&7;
}
static void switch_color_banned(int *color, unsigned long long *seed, int banned) {                       /* :58-64 */
    int combined = *color&banned;
    if (combined == 1 || combined == 2 || combined == 4)
        *color = combined^7;
    else
        switch_color(color, seed);
}
static int symmetrical_trichotomy(int position, int n) { return (int) (3+2.875*position/(n-1)-1.4375+.5)-3; } /* :18-20 */
static int is_corner(v2 aDir, v2 bDir, double crossThreshold) {                                           /* :22-24 */
    return dot(aDir, bDir) <= 0 || fabs(cross(aDir, bDir)) > crossThreshold;
}

/* edgeColoringSimple on one contour (edge-coloring.cpp:73-141); color/seed are the shape-wide running state. edges[] has room for
 * 6 entries at least; corners[] for n. Returns the new edge count. */
static int color_contour_simple(edge_t *edges, int n, int *corners, double crossThreshold, int *color, unsigned long long *seed) {
    if (n == 0)
        return 0;
    int nCorners = 0;
    v2 prevDirection = edge_direction(&edges[n-1], 1);
    for (int i = 0; i < n; ++i) {
        if (is_corner(vnormalize(prevDirection, 0), vnormalize(edge_direction(&edges[i], 0), 0), crossThreshold))
            corners[nCorners++] = i;
        prevDirection = edge_direction(&edges[i], 1);
    }
    if (nCorners == 0) {
        switch_color(color, seed);
        for (int i = 0; i < n; ++i)
            edges[i].color = *color;
    } else if (nCorners == 1) {
        int colors[3];
        switch_color(color, seed);
        colors[0] = *color;
        colors[1] = 7;
        switch_color(color, seed);
        colors[2] = *color;
        int corner = corners[0];
        if (n >= 3) {
            for (int i = 0; i < n; ++i)
                edges[(corner+i)%n].color = colors[1+symmetrical_trichotomy(i, n)];
        } else {
            edge_t parts[6];
            split_in_thirds(&edges[0], parts+3*corner);
            if (n >= 2) {
                split_in_thirds(&edges[1], parts+3-3*corner);
                parts[0].color = parts[1].color = colors[0];
                parts[2].color = parts[3].color = colors[1];
                parts[4].color = parts[5].color = colors[2];
                for (int i = 0; i < 6; ++i)
                    edges[i] = parts[i];
                return 6;
            }
            parts[0].color = colors[0];
            parts[1].color = colors[1];
            parts[2].color = colors[2];
            for (int i = 0; i < 3; ++i)
                edges[i] = parts[i];
            return 3;
        }
    } else {
        int spline = 0, start = corners[0];
        switch_color(color, seed);
        int initialColor = *color;
        for (int i = 0; i < n; ++i) {
            int index = (start+i)%n;
            if (spline+1 < nCorners && corners[spline+1] == index) {
                ++spline;
                switch_color_banned(color, seed, (spline == nCorners-1)*initialColor);
            }
            edges[index].color = *color;
        }
    }
    return n;
}

static double estimate_edge_length(const edge_t *e) {                                         /* edge-coloring.cpp:26-34, MSDFGEN_EDGE_LENGTH_PRECISION 4 */
    double len = 0;
    v2 prev = edge_point(e, 0);
    for (int i = 1; i <= 4; ++i) {
        v2 cur = edge_point(e, 1./4*i);
        len += vlen(vsub(cur, prev));
        prev = cur;
    }
    return len;
}

typedef struct { int index; double prevEdgeLengthEstimate; int minor; int color; } inktrap_corner;   /* edge-coloring.cpp:144-149 */

/* edgeColoringInkTrap on one contour (edge-coloring.cpp:156-257); corners[] has room for n entries. Returns the new edge count. */
static int color_contour_inktrap(edge_t *edges, int n, inktrap_corner *corners, double crossThreshold, int *color, unsigned long long *seed) {
    if (n == 0)
        return 0;
    double splineLength = 0;
    int nCorners = 0;
    v2 prevDirection = edge_direction(&edges[n-1], 1);
    for (int i = 0; i < n; ++i) {
        if (is_corner(vnormalize(prevDirection, 0), vnormalize(edge_direction(&edges[i], 0), 0), crossThreshold)) {
            inktrap_corner c = { i, splineLength, 0, 0 };
            corners[nCorners++] = c;
            splineLength = 0;
        }
        splineLength += estimate_edge_length(&edges[i]);
        prevDirection = edge_direction(&edges[i], 1);
    }
    if (nCorners == 0) {
        switch_color(color, seed);
        for (int i = 0; i < n; ++i)
            edges[i].color = *color;
    } else if (nCorners == 1) {
        int colors[3];
        switch_color(color, seed);
        colors[0] = *color;
        colors[1] = 7;
        switch_color(color, seed);
        colors[2] = *color;
        int corner = corners[0].index;
        if (n >= 3) {
            for (int i = 0; i < n; ++i)
                edges[(corner+i)%n].color = colors[1+symmetrical_trichotomy(i, n)];
        } else {
            edge_t parts[6];
            split_in_thirds(&edges[0], parts+3*corner);
            if (n >= 2) {
                split_in_thirds(&edges[1], parts+3-3*corner);
                parts[0].color = parts[1].color = colors[0];
                parts[2].color = parts[3].color = colors[1];
                parts[4].color = parts[5].color = colors[2];
                for (int i = 0; i < 6; ++i)
                    edges[i] = parts[i];
                return 6;
            }
            parts[0].color = colors[0];
            parts[1].color = colors[1];
            parts[2].color = colors[2];
            for (int i = 0; i < 3; ++i)
                edges[i] = parts[i];
            return 3;
        }
    } else {
        int cornerCount = nCorners, majorCornerCount = nCorners;
        if (cornerCount > 3) {
            corners[0].prevEdgeLengthEstimate += splineLength;
            for (int i = 0; i < cornerCount; ++i) {
                if (corners[i].prevEdgeLengthEstimate > corners[(i+1)%cornerCount].prevEdgeLengthEstimate &&
                    corners[(i+1)%cornerCount].prevEdgeLengthEstimate < corners[(i+2)%cornerCount].prevEdgeLengthEstimate) {
                    corners[i].minor = 1;
                    --majorCornerCount;
                }
            }
        }
        int initialColor = 0;
        for (int i = 0; i < cornerCount; ++i) {
            if (!corners[i].minor) {
                --majorCornerCount;
                switch_color_banned(color, seed, !majorCornerCount*initialColor);
                corners[i].color = *color;
                if (!initialColor)
                    initialColor = *color;
            }
        }
        for (int i = 0; i < cornerCount; ++i) {
            if (corners[i].minor) {
                int nextColor = corners[(i+1)%cornerCount].color;
                corners[i].color = (*color&nextColor)^7;
            } else
                *color = corners[i].color;
        }
        int spline = 0, start = corners[0].index;
        *color = corners[0].color;
        for (int i = 0; i < n; ++i) {
            int index = (start+i)%n;
            if (spline+1 < cornerCount && corners[spline+1].index == index)
                *color = corners[++spline].color;
            edges[index].color = *color;
        }
    }
    return n;
}

/* normalize (if do_normalize) then edgeColoringSimple (coloring == 1) or edgeColoringInkTrap (coloring == 2) of one shape. Output arrays: out_offsets[C+1], out_points[3E*8],
 * out_types[3E], out_colors[3E]. Returns the number of output edges. */
int orc_shape_prepare(const orc_shape *in, int do_normalize, int coloring, double angle_threshold, unsigned long long seed,
                      int32_t *out_offsets, double *out_points, int32_t *out_types, int32_t *out_colors) {
    const int C = in->n_contours;
    double crossThreshold = sin(angle_threshold);
    int color = 0;
    if (coloring == 1 || coloring == 2) {                                                    /* initColor, edge-coloring.cpp:48-51 */
        static const int colors[3] = { 6, 5, 3 };
        color = colors[seed_extract3(&seed)];
    }
    int at = 0;
    out_offsets[0] = 0;
    for (int c = 0; c < C; ++c) {
        const int b = in->contour_offsets[c], n0 = in->contour_offsets[c+1]-b;
        edge_t *edges = (edge_t *) malloc(sizeof(edge_t)*(size_t) (n0 > 2 ? n0 : 6));
        int *corners = (int *) malloc(sizeof(int)*(size_t) (n0 > 2 ? n0 : 6));
        for (int i = 0; i < n0; ++i)
            edges[i] = load_edge(in, b+i);
        int n = n0;
        if (do_normalize)
            n = normalize_contour(edges, n);
        if (coloring == 1)
            n = color_contour_simple(edges, n, corners, crossThreshold, &color, &seed);
        else if (coloring == 2) {
            inktrap_corner *ic = (inktrap_corner *) malloc(sizeof(inktrap_corner)*(size_t) (n > 2 ? n : 6));
            n = color_contour_inktrap(edges, n, ic, crossThreshold, &color, &seed);
            free(ic);
        }
        for (int i = 0; i < n; ++i, ++at) {
            for (int k = 0; k < 4; ++k) {
                out_points[8*(size_t) at+2*k] = k <= edges[i].type ? edges[i].p[k].x : 0;
                out_points[8*(size_t) at+2*k+1] = k <= edges[i].type ? edges[i].p[k].y : 0;
            }
            out_types[at] = edges[i].type;
            out_colors[at] = edges[i].color;
        }
        out_offsets[c+1] = at;
        free(edges);
        free(corners);
    }
    return at;
}

/* ------------------------------------------------------------------------------------------------------------------
 * renderSDF (core/render-sdf.cpp:10-170) and simulate8bit (:172-188). Bitmaps are plain row-major here
 * (the reference's renderSDF does not reorient). Channel pairs as the reference's overloads: out 1 <- sdf 1|3|4 (median of rgb),
 * out 3 <- sdf 1 (replicated) | 3, out 4 <- sdf 4.
 */
static void interpolate_n(float *output, const float *px, int w, int h, int N, v2 pos) {     /* bitmap-interpolation.hpp:10-25 */
    pos.x = dclamp(pos.x, (double) w);
    pos.y = dclamp(pos.y, (double) h);
    pos.x -= .5, pos.y -= .5;
    int l = (int) floor(pos.x), b = (int) floor(pos.y);
    int r = l+1, t = b+1;
    double lr = pos.x-l, bt = pos.y-b;
    l = iclamp(l, w-1), r = iclamp(r, w-1);
    b = iclamp(b, h-1), t = iclamp(t, h-1);
    for (int i = 0; i < N; ++i)
        output[i] = fmix(fmix(px[((size_t) b*w+l)*N+i], px[((size_t) b*w+r)*N+i], lr), fmix(px[((size_t) t*w+l)*N+i], px[((size_t) t*w+r)*N+i], lr), bt);
}

static float dist_val(float dist, double mapScale, double mapTranslate) {                    /* render-sdf.cpp:10-12 */
    double v = mapScale*((double) dist+mapTranslate)+.5;
    return (float) (v >= 0 && v <= 1 ? v : (double) (v > 0));
}

int orc_render_sdf(float *out, int ow, int oh, int No, const float *sdf, int sw, int sh, int Ns, double rangeLower, double rangeUpper, float sdThreshold) {
    if (!((No == 1 && (Ns == 1 || Ns == 3 || Ns == 4)) || (No == 3 && (Ns == 1 || Ns == 3)) || (No == 4 && Ns == 4)))
        return -1;
    v2 scale = V((double) sw/ow, (double) sh/oh);
    const int threshold = rangeLower == rangeUpper;
    double mapScale = 1, mapTranslate = 0;
    float sdBias = 0;
    if (!threshold) {
        double f = (double) (ow+oh)/(sw+sh);                                                 /* Range *= (Range.hpp:20-24) */
        rangeLower *= f, rangeUpper *= f;
        double rangeWidth = rangeUpper-rangeLower;                                           /* DistanceMapping::inverse(Range), DistanceMapping.cpp:6-9 */
        mapScale = rangeWidth, mapTranslate = rangeLower/(rangeWidth ? rangeWidth : 1);
        sdBias = .5f-sdThreshold;
    }
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            float sd[4] = { 0, 0, 0, 0 }, v[4];
            interpolate_n(sd, sdf, sw, sh, Ns, vmulv(scale, V(x+.5, y+.5)));
            int n = No;
            if (No == 1 && Ns >= 3)
                sd[0] = fmedian(sd[0], sd[1], sd[2]);
            if (No == 3 && Ns == 1)
                sd[1] = sd[2] = sd[0];
            for (int i = 0; i < n; ++i)
                v[i] = threshold ? (float) (sd[i] >= sdThreshold) : dist_val(sd[i]+sdBias, mapScale, mapTranslate);
            if (No == 3 && Ns == 1)
                v[1] = v[2] = v[0];
            for (int i = 0; i < n; ++i)
                out[((size_t) y*ow+x)*No+i] = v[i];
        }
    return 0;
}

void orc_simulate_8bit(float *px, long n) {                                                  /* render-sdf.cpp:172-188, pixel-conversion.hpp */
    for (long i = 0; i < n; ++i) {
        unsigned char b;
        orc_pixel_float_to_byte(px+i, &b, 1);
        px[i] = 1.f/255.f*(float) b;
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * estimateSDFError (core/sdf-error-estimation.cpp:134-154) = mean over sub-rows of 1 - overlap(shape scanline, scanline
 * reconstructed from the distance field)/width.
 */
static void scanline_preprocess(scanline_t *line) {                                          /* Scanline.cpp:66-77: sort by x, prefix-sum directions */
    for (int i = 1; i < line->n; ++i) {                                                      /* (ties in x: zero-length spans, order irrelevant to overlap) */
        double x = line->x[i];
        int d = line->dir[i], j = i-1;
        while (j >= 0 && line->x[j] > x) {
            line->x[j+1] = line->x[j], line->dir[j+1] = line->dir[j];
            --j;
        }
        line->x[j+1] = x, line->dir[j+1] = d;
    }
    int total = 0;
    for (int i = 0; i < line->n; ++i) {
        total += line->dir[i];
        line->dir[i] = total;
    }
}

static double scanline_overlap(const scanline_t *a, const scanline_t *b, double xFrom, double xTo, int fillRule) {   /* Scanline.cpp:27-62 */
    double total = 0;
    int aInside = 0, bInside = 0;
    int ai = 0, bi = 0;
    double ax = a->n ? a->x[ai] : xTo;
    double bx = b->n ? b->x[bi] : xTo;
    while (ax < xFrom || bx < xFrom) {
        double xNext = dmin(ax, bx);
        if (ax == xNext && ai < a->n) {
            aInside = interpret_fill_rule(a->dir[ai], fillRule);
            ax = ++ai < a->n ? a->x[ai] : xTo;
        }
        if (bx == xNext && bi < b->n) {
            bInside = interpret_fill_rule(b->dir[bi], fillRule);
            bx = ++bi < b->n ? b->x[bi] : xTo;
        }
    }
    double x = xFrom;
    while (ax < xTo || bx < xTo) {
        double xNext = dmin(ax, bx);
        if (aInside == bInside)
            total += xNext-x;
        if (ax == xNext && ai < a->n) {
            aInside = interpret_fill_rule(a->dir[ai], fillRule);
            ax = ++ai < a->n ? a->x[ai] : xTo;
        }
        if (bx == xNext && bi < b->n) {
            bInside = interpret_fill_rule(b->dir[bi], fillRule);
            bx = ++bi < b->n ? b->x[bi] : xTo;
        }
        x = xNext;
    }
    if (aInside == bInside)
        total += xTo-x;
    return total;
}

/* scanlineSDF (N == 1, sdf-error-estimation.cpp:9-48) and scanlineMSDF (N >= 3, :50-125). px: row-major [h][w][N] memory rows;
 * yDownBitmap: the shape's Y axis points down (the reference passes shape.getYAxisOrientation(), :146). */
static void scanline_from_sdf(scanline_t *line, const float *px, int w, int h, int N, double sy, double ty, double sx, double tx, double y, int yDown) {
    line->n = 0;
    if (!(w > 0 && h > 0))
        return;
    double pixelY = dclamp(sy*(y+ty)-.5, (double) (h-1));                                    /* projection.projectY(y), Projection.cpp:30-32 */
    if (yDown)
        pixelY = h-1-pixelY;
    int b = (int) floor(pixelY);
    int t = b+1;
    double bt = pixelY-b;
    if (t >= h) {
        b = h-1;
        t = h-1;
        bt = 1;
    }
    int inside = 0;
    #define SDF_AT(X, Y, C) px[((size_t) (Y)*w+(X))*N+(C)]
    if (N == 1) {
        float lv, rv = fmix(SDF_AT(0, b, 0), SDF_AT(0, t, 0), bt);
        if ((inside = rv > .5f)) {
            line->x[line->n] = -1e240, line->dir[line->n++] = 1;
        }
        for (int l = 0, r = 1; r < w; ++l, ++r) {
            lv = rv;
            rv = fmix(SDF_AT(r, b, 0), SDF_AT(r, t, 0), bt);
            if (lv != rv) {
                double lr = (double) (.5f-lv)/(double) (rv-lv);
                if (lr >= 0 && lr <= 1) {
                    line->x[line->n] = (l+lr+.5)/sx-tx;                                      /* projection.unprojectX, Projection.cpp:34-36 */
                    line->dir[line->n++] = isign(rv-lv);
                }
            }
        }
    } else {
        float lv[3], rv[3];
        for (int i = 0; i < 3; ++i)
            rv[i] = fmix(SDF_AT(0, b, i), SDF_AT(0, t, i), bt);
        if ((inside = fmedian(rv[0], rv[1], rv[2]) > .5f)) {
            line->x[line->n] = -1e240, line->dir[line->n++] = 1;
        }
        for (int l = 0, r = 1; r < w; ++l, ++r) {
            for (int i = 0; i < 3; ++i) {
                lv[i] = rv[i];
                rv[i] = fmix(SDF_AT(r, b, i), SDF_AT(r, t, i), bt);
            }
            double nx[4];
            int nd[4], count = 0;
            for (int i = 0; i < 3; ++i) {
                if (lv[i] != rv[i]) {
                    double lr = (double) (.5f-lv[i])/(double) (rv[i]-lv[i]);
                    if (lr >= 0 && lr <= 1) {
                        float v[3] = { fmix(lv[0], rv[0], lr), fmix(lv[1], rv[1], lr), fmix(lv[2], rv[2], lr) };
                        if (fmedian(v[0], v[1], v[2]) == v[i]) {
                            nx[count] = (l+lr+.5)/sx-tx;
                            nd[count] = isign(rv[i]-lv[i]);
                            ++count;
                        }
                    }
                }
            }
            #define SWAP_NI(A, B) { nx[3] = nx[A], nd[3] = nd[A]; nx[A] = nx[B], nd[A] = nd[B]; nx[B] = nx[3], nd[B] = nd[3]; }
            if (count >= 2) {
                if (nx[0] > nx[1])
                    SWAP_NI(0, 1)
                if (count >= 3 && nx[1] > nx[2]) {
                    SWAP_NI(1, 2)
                    if (nx[0] > nx[1])
                        SWAP_NI(0, 1)
                }
            }
            #undef SWAP_NI
            for (int i = 0; i < count; ++i) {
                if ((nd[i] > 0) == !inside) {
                    line->x[line->n] = nx[i], line->dir[line->n++] = nd[i];
                    inside = !inside;
                }
            }
            float rvScalar = fmedian(rv[0], rv[1], rv[2]);
            if ((rvScalar > .5f) != inside && rvScalar != .5f && line->n > 0) {
                --line->n;
                inside = !inside;
            }
        }
    }
    #undef SDF_AT
    scanline_preprocess(line);
}

/* per_line (optional): (h-1)*scanlinesPerRow values 1 - overlapFactor*overlap in the order they are summed. */
double orc_estimate_sdf_error(const orc_shape *shape, const float *px, int w, int h, int N, const double *xf4, int scanlinesPerRow, int fillRule,
                              double *per_line) {
    if (w <= 1 || h <= 1 || scanlinesPerRow < 1)
        return 0;
    const double sx = xf4[0], sy = xf4[1], tx = xf4[2], ty = xf4[3];
    double subRowSize = 1./scanlinesPerRow;
    double xFrom = .5/sx-tx;
    double xTo = (w-.5)/sx-tx;
    double overlapFactor = 1/(xTo-xFrom);
    double error = 0;
    int nE = shape->contour_offsets[shape->n_contours];
    scanline_t ref, sdf;
    ref.cap = 3*nE+1, sdf.cap = 3*w+2;
    ref.x = (double *) malloc(sizeof(double)*(size_t) ref.cap), ref.dir = (int *) malloc(sizeof(int)*(size_t) ref.cap);
    sdf.x = (double *) malloc(sizeof(double)*(size_t) sdf.cap), sdf.dir = (int *) malloc(sizeof(int)*(size_t) sdf.cap);
    for (int row = 0; row < h-1; ++row)
        for (int subRow = 0; subRow < scanlinesPerRow; ++subRow) {
            double bt = (subRow+.5)*subRowSize;
            double y = (row+bt+.5)/sy-ty;
            shape_scanline(shape, &ref, y);
            scanline_preprocess(&ref);
            scanline_from_sdf(&sdf, px, w, h, N, sy, ty, sx, tx, y, shape->inverse_y);
            double v = 1-overlapFactor*scanline_overlap(&ref, &sdf, xFrom, xTo, fillRule);
            if (per_line)
                per_line[(size_t) row*scanlinesPerRow+subRow] = v;
            error += v;
        }
    free(ref.x), free(ref.dir), free(sdf.x), free(sdf.dir);
    return error/((h-1)*scanlinesPerRow);
}
