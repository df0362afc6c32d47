// include/key_value_store.h -- shared machinery of ParameterFile (par_file.h) and ParameterArgs (par_args.h).
//
// Both of the reference's option sources (src/include/par_file.h, src/include/par_args.h) are a string -> string map
// with the same typed getters: a value is parsed with `istringstream >> T` (so a string is its first blank-delimited
// token, a bool is 0 / 1, trailing characters are ignored) and a missing or unparsable key throws.  This header holds
// that map once; the two front ends differ only in how they fill it and in their error wording.
#ifndef KR_KEY_VALUE_STORE_H_
#define KR_KEY_VALUE_STORE_H_

#include <exception>
#include <map>
#include <sstream>
#include <string>
#include <typeinfo>

namespace krhost {

class OptionError : public std::exception {
public:
    explicit OptionError(std::string text) : text_(std::move(text)) {}
    const char* what() const noexcept override { return text_.c_str(); }

private:
    std::string text_;
};

inline std::string strip_blanks(const std::string& s)
{
    const std::size_t b = s.find_first_not_of("\t ");
    if (b == std::string::npos) return std::string();
    return s.substr(b, s.find_last_not_of("\t ") - b + 1);
}

// "key = value" -> (key, value), both stripped; false when there is no '=' or nothing before it
inline bool split_assignment(const std::string& line, std::string& key, std::string& value)
{
    const std::size_t eq = line.find('=');
    if (eq == std::string::npos) return false;
    key = strip_blanks(line.substr(0, eq));
    value = strip_blanks(line.substr(eq + 1));
    return !key.empty();
}

template <typename Error>
class KeyValueStore {
public:
    bool key_exists(const std::string& key) const { return entries_.count(key) != 0; }

    template <typename T>
    T get_parameter(const std::string& key) const
    {
        return convert<T>(key, lookup(key));
    }

    template <typename T>
    T get_parameter(const std::string& key, T fallback) const
    {
        const auto it = entries_.find(key);
        return it == entries_.end() ? fallback : convert<T>(key, it->second);
    }

    template <typename T>
    void get_parameter_array(const std::string& key, T* out, int count) const
    {
        std::istringstream in(lookup(key));
        for (int i = 0; i < count; ++i) {
            if (!(in >> out[i])) {
                std::ostringstream msg;
                msg << "Could not parse array " << key << " (expected " << count << " values of type " << typeid(T).name() << ")";
                throw Error(msg.str());
            }
        }
    }

    template <typename T>
    static T string_to_T(const std::string& text)
    {
        std::istringstream in(text);
        T v;
        if (!(in >> v)) return T();
        return v;
    }

protected:
    // false (and nothing stored) when the key is already present: the caller decides whether that is fatal
    bool store(const std::string& key, const std::string& value) { return entries_.emplace(key, value).second; }

    const std::string& lookup(const std::string& key) const
    {
        const auto it = entries_.find(key);
        if (it == entries_.end()) throw Error(missing_text(key));
        return it->second;
    }

    virtual std::string missing_text(const std::string& key) const = 0;
    virtual ~KeyValueStore() = default;

private:
    template <typename T>
    T convert(const std::string& key, const std::string& text) const
    {
        std::istringstream in(text);
        T v;
        if (!(in >> v)) throw Error("Could not parse value of " + key + " (expected type " + typeid(T).name() + ")");
        return v;
    }

    std::map<std::string, std::string> entries_;
};

}   // namespace krhost

#endif /* KR_KEY_VALUE_STORE_H_ */
